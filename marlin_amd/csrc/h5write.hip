// Minimal HDF5 container writer (host code only): what XDMFTensorOutput needs from libhdf5 -- H5Fcreate / one H5Dcreate + H5Dwrite per
// "<buffer>.<frame>" dataset in the root group / H5Fflush / H5Fclose (src/tensor_outputs/XDMFTensorOutput.C:152-160, 244-246,
// 578-650) -- written against the published HDF5 file format specification (version 0 superblock, version 1 object headers, a
// symbol-table root group), because the image has no libhdf5 to link.  The files are ordinary HDF5: h5dump / h5py / ParaView's
// XDMF reader open them, and the reference's HDF5Diff tester (scripts/TestHarness/testers/HDF5Diff.py:14-87: same dataset names and
// shapes, max |a - b| <= abs_tol) compares them against its gold files.  Differences from the reference's files that a reader
// cannot observe through the dataset API: contiguous instead of chunked + deflate storage, no modification times.
//
// File layout: [superblock 96 B][dataset data, 8-byte aligned, appended as it arrives] ... [metadata block].  Every flush writes a
// fresh metadata block (dataset object headers, local heap of names, symbol-table nodes, one B-tree node, the root object header) at
// the end of the file and then repoints the superblock at it, so the file on disk is valid after every frame, as after the
// reference's H5Fflush; data that arrives later is appended behind the (then dead) block.
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

#include "marlin_hip.h"

#define MRL_TRY(expr)              \
  do {                             \
    int rc_ = (expr);              \
    if (rc_ != MRL_OK) return rc_; \
  } while (0)

namespace {

constexpr uint64_t kUndef = ~0ULL;
constexpr int kLeafK = 32;       // symbols per symbol-table node: 2 * kLeafK
constexpr int kInternalK = 256;  // children of the (single, level-0) B-tree node: 2 * kInternalK

struct Dataset {
  std::string name;
  int dtype, rank;
  uint64_t dims[4];
  uint64_t addr, bytes;
  uint64_t oh = 0;   // address of the dataset's object header: written ONCE, right behind its data (it never changes)
};

struct Buf {
  std::vector<unsigned char> b;
  void u8(unsigned v) { b.push_back((unsigned char)v); }
  void u16(unsigned v) { u8(v & 255); u8((v >> 8) & 255); }
  void u32(uint32_t v) { for (int i = 0; i < 4; ++i) u8((v >> (8 * i)) & 255); }
  void u64(uint64_t v) { for (int i = 0; i < 8; ++i) u8((unsigned)((v >> (8 * i)) & 255)); }
  void raw(const void *p, size_t n) { const unsigned char *c = static_cast<const unsigned char *>(p); b.insert(b.end(), c, c + n); }
  void pad8() { while (b.size() % 8) u8(0); }
  void zeros(size_t n) { b.insert(b.end(), n, 0); }
  size_t size() const { return b.size(); }
};

size_t type_size(int dtype) {
  switch (dtype) {
    case MRL_H5_F64: case MRL_H5_I64: return 8;
    case MRL_H5_F32: case MRL_H5_I32: return 4;
  }
  return 0;
}

// one header message: type, size of the (8-byte padded) body, flags
void message(Buf &o, unsigned type, const Buf &body, unsigned flags = 0) {
  Buf p = body;
  p.pad8();
  o.u16(type);
  o.u16((unsigned)p.size());
  o.u8(flags);
  o.u8(0); o.u8(0); o.u8(0);
  o.raw(p.b.data(), p.size());
}

// version 1 object header around `msgs` (nmsg messages)
Buf object_header(const Buf &msgs, int nmsg) {
  Buf o;
  o.u8(1);                       // version
  o.u8(0);
  o.u16((unsigned)nmsg);
  o.u32(1);                      // object reference count
  o.u32((uint32_t)msgs.size());  // header size (message bytes)
  o.u32(0);                      // pad to 8
  o.raw(msgs.b.data(), msgs.size());
  return o;
}

Buf dataset_header(const Dataset &d) {
  Buf msgs;
  {  // dataspace, version 1
    Buf m;
    m.u8(1); m.u8((unsigned)d.rank); m.u8(0); m.u8(0); m.u32(0);
    for (int i = 0; i < d.rank; ++i) m.u64(d.dims[i]);
    message(msgs, 0x0001, m);
  }
  {  // datatype, version 1
    Buf m;
    const unsigned sz = (unsigned)type_size(d.dtype);
    if (d.dtype == MRL_H5_F64 || d.dtype == MRL_H5_F32) {
      const bool dbl = d.dtype == MRL_H5_F64;
      m.u8(0x11);                 // class 1 (floating point), version 1
      m.u8(0x20);                 // little endian, mantissa normalisation: implied leading 1
      m.u8(dbl ? 63 : 31);        // sign bit
      m.u8(0);
      m.u32(sz);
      m.u16(0); m.u16(8 * sz);    // bit offset, precision
      m.u8(dbl ? 52 : 23); m.u8(dbl ? 11 : 8);   // exponent location, size
      m.u8(0); m.u8(dbl ? 52 : 23);              // mantissa location, size
      m.u32(dbl ? 1023 : 127);    // exponent bias
    } else {
      m.u8(0x10);                 // class 0 (fixed point), version 1
      m.u8(0x08);                 // little endian, two's complement signed
      m.u8(0); m.u8(0);
      m.u32(sz);
      m.u16(0); m.u16(8 * sz);
    }
    message(msgs, 0x0003, m, 1);
  }
  {  // fill value, version 2: allocate late, write if set, default fill value
    Buf m;
    m.u8(2); m.u8(2); m.u8(2); m.u8(1); m.u32(0);
    message(msgs, 0x0005, m, 1);
  }
  {  // data layout, version 3: contiguous
    Buf m;
    m.u8(3); m.u8(1); m.u64(d.addr); m.u64(d.bytes);
    message(msgs, 0x0008, m);
  }
  return object_header(msgs, 4);
}

}  // namespace

struct mrl_h5 {
  FILE *f = nullptr;
  std::string path, err;
  std::vector<Dataset> sets;
  uint64_t eod = 96;   // end of the dataset data written so far (a metadata block may follow it on disk)
  bool dirty = true;
};

namespace {

int fail(mrl_h5 *h, int code, const std::string &msg) {
  h->err = msg;
  return code;
}

int write_at(mrl_h5 *h, uint64_t at, const void *p, size_t n) {
  if (fseeko(h->f, (off_t)at, SEEK_SET) != 0 || (n && fwrite(p, 1, n, h->f) != n)) return fail(h, MRL_ERR_IO, "write to " + h->path + " failed");
  return MRL_OK;
}

int write_metadata(mrl_h5 *h) {
  // names in the order the group B-tree requires (strcmp)
  std::vector<int> order(h->sets.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
  std::sort(order.begin(), order.end(), [&](int a, int b) { return std::strcmp(h->sets[a].name.c_str(), h->sets[b].name.c_str()) < 0; });
  const size_t n = order.size(), per = 2 * kLeafK, nsnod = (n + per - 1) / per;   // (an empty group: a B-tree node without entries)
  if (nsnod > 2 * (size_t)kInternalK) return fail(h, MRL_ERR_UNSUPPORTED, "more than 32768 datasets in one HDF5 file");
  uint64_t at = (h->eod + 7) & ~7ULL;
  const uint64_t meta0 = at;
  Buf m;
  // dataset object headers: each was written once, behind its data, by mrl_h5_write.  A flush re-appends only what DOES change with
  // every new dataset -- the name heap, the symbol-table nodes, the B-tree node and the root header -- so the dead metadata a flush per
  // output step leaves behind is ~50 B per dataset and flush plus 8 KB per flush instead of ~400 B per dataset and flush
  std::vector<uint64_t> oh(n);
  for (size_t i = 0; i < n; ++i) oh[i] = h->sets[order[i]].oh;
  // local heap: header + data segment (offset 0 = the empty name of the root group)
  Buf names;
  names.zeros(8);
  std::vector<uint64_t> noff(n);
  for (size_t i = 0; i < n; ++i) {
    noff[i] = names.size();
    const std::string &s = h->sets[order[i]].name;
    names.raw(s.c_str(), s.size() + 1);
    names.pad8();
  }
  const uint64_t heap_addr = meta0 + m.size();
  m.raw("HEAP", 4);
  m.u8(0); m.u8(0); m.u8(0); m.u8(0);
  m.u64(names.size());
  m.u64(1);                           // no free block (H5HL_FREE_NULL)
  m.u64(heap_addr + 32);              // data segment follows the header
  m.raw(names.b.data(), names.size());
  // symbol-table nodes
  std::vector<uint64_t> snod(nsnod), last_name(nsnod, 0);
  for (size_t s = 0; s < nsnod; ++s) {
    snod[s] = meta0 + m.size();
    const size_t lo = s * per, hi = std::min(n, lo + per);
    m.raw("SNOD", 4);
    m.u8(1); m.u8(0);
    m.u16((unsigned)(hi > lo ? hi - lo : 0));
    for (size_t i = lo; i < lo + per; ++i) {
      if (i < hi) {
        m.u64(noff[i]); m.u64(oh[i]); m.u32(0); m.u32(0); m.zeros(16);
        last_name[s] = noff[i];
      } else {
        m.zeros(40);
      }
    }
  }
  // the B-tree node (level 0) over them
  const uint64_t btree_addr = meta0 + m.size();
  m.raw("TREE", 4);
  m.u8(0); m.u8(0);
  m.u16((unsigned)nsnod);
  m.u64(kUndef); m.u64(kUndef);
  m.u64(0);                           // key 0: the empty string
  for (size_t s = 0; s < 2 * (size_t)kInternalK; ++s) {
    if (s < nsnod) { m.u64(snod[s]); m.u64(last_name[s]); } else { m.u64(0); m.u64(0); }
  }
  // root group object header: one symbol-table message
  const uint64_t root_addr = meta0 + m.size();
  {
    Buf msgs, st;
    st.u64(btree_addr); st.u64(heap_addr);
    message(msgs, 0x0011, st);
    Buf r = object_header(msgs, 1);
    m.raw(r.b.data(), r.size());
    m.pad8();
  }
  const uint64_t eof = meta0 + m.size();
  // superblock, version 0
  Buf sb;
  const unsigned char sig[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
  sb.raw(sig, 8);
  sb.u8(0); sb.u8(0); sb.u8(0); sb.u8(0); sb.u8(0);
  sb.u8(8); sb.u8(8); sb.u8(0);
  sb.u16(kLeafK); sb.u16(kInternalK);
  sb.u32(0);
  sb.u64(0); sb.u64(kUndef); sb.u64(eof); sb.u64(kUndef);
  sb.u64(0); sb.u64(root_addr); sb.u32(1); sb.u32(0); sb.u64(btree_addr); sb.u64(heap_addr);   // root symbol-table entry
  if (h->eod < meta0) {  // alignment gap
    const unsigned char z[8] = {0};
    MRL_TRY(write_at(h, h->eod, z, (size_t)(meta0 - h->eod)));
  }
  // order: the new metadata block reaches the file before the superblock that points at it
  MRL_TRY(write_at(h, meta0, m.b.data(), m.size()));
  if (fflush(h->f) != 0) return fail(h, MRL_ERR_IO, "flush of " + h->path + " failed");
  MRL_TRY(write_at(h, 0, sb.b.data(), sb.size()));
  if (fflush(h->f) != 0) return fail(h, MRL_ERR_IO, "flush of " + h->path + " failed");
  // the block just written is LIVE (the superblock on disk points at it): the next dataset goes behind it, so the file stays valid
  // and complete up to the last flush at every moment in between (H5Fflush semantics, XDMFTensorOutput.C:244-246)
  h->eod = eof;
  h->dirty = false;
  return MRL_OK;
}

}  // namespace

extern "C" {

int mrl_h5_create(const char *path, mrl_h5 **out) {
  if (!path || !out) return MRL_ERR_INVALID;
  *out = nullptr;
  FILE *f = std::fopen(path, "wb+");
  if (!f) return MRL_ERR_IO;
  mrl_h5 *h = new mrl_h5();
  h->f = f;
  h->path = path;
  const int rc = write_metadata(h);   // an empty but valid file (H5Fcreate)
  if (rc != MRL_OK) {
    std::fclose(f);
    delete h;
    return rc;
  }
  *out = h;
  return MRL_OK;
}

int mrl_h5_write(mrl_h5 *h, const char *name, int dtype, int rank, const int64_t *dims, const void *data) {
  if (!h || !h->f) return MRL_ERR_INVALID;
  if (!name || !*name || std::strchr(name, '/')) return fail(h, MRL_ERR_INVALID, "mrl_h5_write: dataset names are plain link names in the root group");
  if (rank < 1 || rank > 4 || !dims || !data || !type_size(dtype)) return fail(h, MRL_ERR_INVALID, "mrl_h5_write: bad rank / dims / dtype / data");
  for (const Dataset &d : h->sets)
    if (d.name == name) return fail(h, MRL_ERR_INVALID, std::string("Dataset '") + name + "' already exists in HDF5 file.");   // :593-594
  Dataset d{};
  d.name = name;
  d.dtype = dtype;
  d.rank = rank;
  uint64_t count = 1;
  for (int i = 0; i < rank; ++i) {
    if (dims[i] <= 0) return fail(h, MRL_ERR_INVALID, "mrl_h5_write: dimensions must be positive");
    d.dims[i] = (uint64_t)dims[i];
    count *= d.dims[i];
  }
  d.bytes = count * type_size(dtype);
  d.addr = (h->eod + 7) & ~7ULL;
  if (d.addr > h->eod) {
    const unsigned char z[8] = {0};
    MRL_TRY(write_at(h, h->eod, z, (size_t)(d.addr - h->eod)));
  }
  MRL_TRY(write_at(h, d.addr, data, (size_t)d.bytes));
  h->eod = d.addr + d.bytes;
  {  // its object header, once and for all (nothing points at it until the next flush publishes a symbol-table entry)
    d.oh = (h->eod + 7) & ~7ULL;
    if (d.oh > h->eod) {
      const unsigned char z[8] = {0};
      MRL_TRY(write_at(h, h->eod, z, (size_t)(d.oh - h->eod)));
    }
    Buf hd = dataset_header(d);
    hd.pad8();
    MRL_TRY(write_at(h, d.oh, hd.b.data(), hd.size()));
    h->eod = d.oh + hd.size();
  }
  h->sets.push_back(d);
  h->dirty = true;
  return MRL_OK;
}

int mrl_h5_flush(mrl_h5 *h) {
  if (!h || !h->f) return MRL_ERR_INVALID;
  return h->dirty ? write_metadata(h) : MRL_OK;
}

int mrl_h5_close(mrl_h5 *h) {
  if (!h) return MRL_ERR_INVALID;
  int rc = MRL_OK;
  if (h->f) {
    if (h->dirty) rc = write_metadata(h);
    if (std::fclose(h->f) != 0 && rc == MRL_OK) rc = MRL_ERR_IO;
  }
  delete h;
  return rc;
}

const char *mrl_h5_last_error(const mrl_h5 *h) { return h ? h->err.c_str() : "null handle"; }

}  // extern "C"
