// Power-of-two Stockham FFT building blocks for gfx950 (fp64, no MFMA: the path is HBM-bound).
//
// Every thread owns 16 complex values of one line: positions q + m*(N/16), m = 0..15, where
// q = thread index within the line (N/16 threads per line).  A transform is 2 or 3 radix stages
// (16/8/4 point butterflies in registers); between stages the line is exchanged through LDS in
// natural (Stockham autosort) order, after which each thread again owns q + m*(N/16).  The first
// stage therefore loads straight from HBM and the last stage stores straight to HBM with the same
// index pattern, and 256-thread workgroups always hold 4096 points = 64 KiB of exchange space.
// Twiddles exp(-2 pi i k/N) are staged once per workgroup into LDS (exact table values, no
// recurrences).  Inverse transforms use the swap trick: ifft(x) = swap(fft(swap(x))).
#pragma once
#include "mrl_internal.h"

namespace mrl {
namespace p2 {

__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ cplx mul_mi(cplx a) { return make_double2(a.y, -a.x); }  // * (-i)
__device__ __forceinline__ cplx cswap(cplx a) { return make_double2(a.y, a.x); }

template <int N>
struct Plan;
template <>
struct Plan<64> {
  static constexpr int ns = 2;
  static constexpr int r0 = 8, r1 = 8, r2 = 1;
};
template <>
struct Plan<128> {
  static constexpr int ns = 2;
  static constexpr int r0 = 16, r1 = 8, r2 = 1;
};
template <>
struct Plan<256> {
  static constexpr int ns = 2;
  static constexpr int r0 = 16, r1 = 16, r2 = 1;
};
template <>
struct Plan<512> {
  static constexpr int ns = 3;
  static constexpr int r0 = 8, r1 = 8, r2 = 8;
};
template <>
struct Plan<1024> {
  static constexpr int ns = 3;
  static constexpr int r0 = 16, r1 = 8, r2 = 8;
};

__device__ __forceinline__ void bfly4(cplx &a0, cplx &a1, cplx &a2, cplx &a3) {
  const cplx t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = mul_mi(csub(a1, a3));
  a0 = cadd(t0, t2);
  a1 = cadd(t1, t3);
  a2 = csub(t0, t2);
  a3 = csub(t1, t3);
}

// natural-order DFT of R register values a[0..R-1]
template <int R>
__device__ __forceinline__ void bfly(cplx (&a)[R]);

template <>
__device__ __forceinline__ void bfly<4>(cplx (&a)[4]) {
  bfly4(a[0], a[1], a[2], a[3]);
}

template <>
__device__ __forceinline__ void bfly<8>(cplx (&a)[8]) {
  const double h = 0.70710678118654752440;
  cplx e0 = a[0], e1 = a[2], e2 = a[4], e3 = a[6];
  cplx o0 = a[1], o1 = a[3], o2 = a[5], o3 = a[7];
  bfly4(e0, e1, e2, e3);
  bfly4(o0, o1, o2, o3);
  o1 = make_double2(h * (o1.x + o1.y), h * (o1.y - o1.x));    // * W8^1 = (h, -h)
  o2 = mul_mi(o2);                                            // * W8^2 = -i
  o3 = make_double2(h * (o3.y - o3.x), -h * (o3.x + o3.y));   // * W8^3 = (-h, -h)
  a[0] = cadd(e0, o0);
  a[4] = csub(e0, o0);
  a[1] = cadd(e1, o1);
  a[5] = csub(e1, o1);
  a[2] = cadd(e2, o2);
  a[6] = csub(e2, o2);
  a[3] = cadd(e3, o3);
  a[7] = csub(e3, o3);
}

template <>
__device__ __forceinline__ void bfly<16>(cplx (&a)[16]) {
  const double c1 = 0.92387953251128675613, s1 = 0.38268343236508977173, h = 0.70710678118654752440;
  // inner DFT-4 over n1 for each n2: (a[n2], a[n2+4], a[n2+8], a[n2+12]) -> A[n2][k1] at a[n2+4*k1]
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) bfly4(a[n2], a[n2 + 4], a[n2 + 8], a[n2 + 12]);
  // twiddles W16^(n2*k1)
  a[5] = cmul(a[5], make_double2(c1, -s1));                      // 1*1
  a[9] = make_double2(h * (a[9].x + a[9].y), h * (a[9].y - a[9].x));  // 1*2 -> W16^2
  a[13] = cmul(a[13], make_double2(s1, -c1));                    // 1*3
  a[6] = make_double2(h * (a[6].x + a[6].y), h * (a[6].y - a[6].x));  // 2*1 -> W16^2
  a[10] = mul_mi(a[10]);                                         // 2*2 -> W16^4
  a[14] = make_double2(h * (a[14].y - a[14].x), -h * (a[14].x + a[14].y));  // 2*3 -> W16^6
  a[7] = cmul(a[7], make_double2(s1, -c1));                      // 3*1 -> W16^3
  a[11] = make_double2(h * (a[11].y - a[11].x), -h * (a[11].x + a[11].y));  // 3*2 -> W16^6
  a[15] = cmul(a[15], make_double2(-c1, s1));                    // 3*3 -> W16^9
  // outer DFT-4 over n2 for each k1: X[k1 + 4*k2] lands at a[4*k1 + k2]
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) bfly4(a[4 * k1], a[4 * k1 + 1], a[4 * k1 + 2], a[4 * k1 + 3]);
  // transpose to natural order
  cplx t;
#define MRL_SWAP(i, j) \
  t = a[i];            \
  a[i] = a[j];         \
  a[j] = t;
  MRL_SWAP(1, 4) MRL_SWAP(2, 8) MRL_SWAP(3, 12) MRL_SWAP(6, 9) MRL_SWAP(7, 13) MRL_SWAP(11, 14)
#undef MRL_SWAP
}

// LDS index maps: p = position within the line, l = line within the workgroup
template <int N>
struct MapStrided {  // lines fastest (lanes of a wave vary l): conflict-free without padding
  static constexpr int T = 4096 / N;
  __device__ __forceinline__ static int at(int p, int l) { return p * T + l; }
  static constexpr int size = 4096;
};
template <int N>
struct MapLine {  // position fastest (lanes vary q): one pad element per 16 positions
  static constexpr int LP = N + N / 16;
  __device__ __forceinline__ static int at(int p, int l) { return l * LP + p + (p >> 4); }
  static constexpr int size = (4096 / N) * LP;
};

// radix stage STAGE on the 16 register values (v[i + S*t] = element t of butterfly i)
template <int N, int R, int NS>
__device__ __forceinline__ void stage(cplx (&v)[16], int q, const cplx *W) {
  constexpr int S = 16 / R;
#pragma unroll
  for (int i = 0; i < S; ++i) {
    cplx a[R];
#pragma unroll
    for (int t = 0; t < R; ++t) a[t] = v[i + S * t];
    if (NS > 1) {
      const int b = q + i * (N / 16);
      const int k = b % NS;
      const int step = k * (N / (NS * R));
#pragma unroll
      for (int t = 1; t < R; ++t) a[t] = cmul(a[t], W[t * step]);
    }
    bfly<R>(a);
#pragma unroll
    for (int t = 0; t < R; ++t) v[i + S * t] = a[t];
  }
}

// write the outputs of a radix-R stage (Ns = NS) in Stockham order, then re-own q + m*N/16
template <int N, int R, int NS, class Map>
__device__ __forceinline__ void exchange(cplx (&v)[16], int q, int l, cplx *X) {
  constexpr int S = 16 / R;
  __syncthreads();  // previous readers of X are done
#pragma unroll
  for (int i = 0; i < S; ++i) {
    const int b = q + i * (N / 16);
    const int p0 = (b / NS) * NS * R + (b % NS);
#pragma unroll
    for (int t = 0; t < R; ++t) X[Map::at(p0 + t * NS, l)] = v[i + S * t];
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 16; ++m) v[m] = X[Map::at(q + m * (N / 16), l)];
}

// full forward transform of the line owned by (q, l); v in: x[q + m*N/16], out: X[q + m*N/16]
template <int N, class Map>
__device__ __forceinline__ void fft_line(cplx (&v)[16], int q, int l, cplx *X, const cplx *W) {
  using P = Plan<N>;
  stage<N, P::r0, 1>(v, q, W);
  exchange<N, P::r0, 1, Map>(v, q, l, X);
  stage<N, P::r1, P::r0>(v, q, W);
  if (P::ns == 3) {
    exchange<N, P::r1, P::r0, Map>(v, q, l, X);
    stage<N, P::r2, P::r0 * P::r1>(v, q, W);
  }
}

// bijective XCD-aware remap: hardware deals block b to XCD b % 8; give each XCD a contiguous
// range of logical tiles so neighbouring tiles (which share partial 128-B lines) share an L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nb) {
  const unsigned q8 = nb >> 3, r8 = nb & 7, xcd = b & 7, idx = b >> 3;
  return (xcd < r8) ? xcd * (q8 + 1) + idx : r8 * (q8 + 1) + (xcd - r8) * q8 + idx;
}

// the same ranges walked in descending order: a consumer kernel that starts where its producer ended finds the
// producer's last ~256 MB still in the Infinity Cache
__device__ __forceinline__ unsigned xcd_remap_rev(unsigned b, unsigned nb) {
  const unsigned q8 = nb >> 3, r8 = nb & 7, xcd = b & 7, idx = b >> 3;
  const unsigned len = q8 + (xcd < r8 ? 1u : 0u);
  const unsigned start = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  return start + (len - 1 - idx);
}

template <int N>
__device__ __forceinline__ void load_twiddles(cplx *W, const cplx *__restrict__ tw) {
  for (int i = threadIdx.x; i < N; i += 256) W[i] = tw[i];
}

// Twiddle staging split in two so that a kernel can issue the table loads BEFORE its operand loads (vmcnt
// retires in order: the small L2-resident table loads must not queue behind a full HBM round trip) and
// write them to LDS AFTER the operand loads are in flight.
template <int N>
struct TwRegs {
  cplx v[(N + 255) / 256];
};
template <int N>
__device__ __forceinline__ void tw_issue(TwRegs<N> &r, const cplx *__restrict__ tw) {
#pragma unroll
  for (int j = 0; j < (N + 255) / 256; ++j) {
    const int idx = threadIdx.x + j * 256;
    r.v[j] = idx < N ? tw[idx] : make_double2(0.0, 0.0);
  }
}
template <int N>
__device__ __forceinline__ void tw_commit(const TwRegs<N> &r, cplx *W) {
#pragma unroll
  for (int j = 0; j < (N + 255) / 256; ++j) {
    const int idx = threadIdx.x + j * 256;
    if (idx < N) W[idx] = r.v[j];
  }
}

}  // namespace p2
}  // namespace mrl
