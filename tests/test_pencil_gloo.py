"""world_size-4 / 6 gloo runs (CPU) of the FFT_PENCIL exchange pattern with REAL messages: every rank transforms its block with torch,
packs the messages of the two forward stages (DomainAction.C:1105-1256) by the LIBRARY's host-side layout (mrl_pencil_layout: the
function the HIP pipeline sizes its own exchange buffers with), sends them with all_to_all_single over gloo, unpacks -- and must end
with its block of the serial transform of the global array; then the inverse stages (:1258-1404) back to its real block."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _exchange(send_chunks, send_counts, recv_counts):
    """all_to_all_single of complex chunks (as interleaved doubles) with per-peer element counts"""
    send = torch.cat([c.reshape(-1) for c in send_chunks]) if send_chunks else torch.zeros(0, dtype=torch.complex128)
    assert [c.numel() for c in send_chunks] == send_counts
    sr = torch.view_as_real(send.contiguous()).reshape(-1)
    rr = torch.empty(2 * sum(recv_counts), dtype=torch.float64)
    dist.all_to_all_single(rr, sr, [2 * c for c in recv_counts], [2 * c for c in send_counts])
    flat = torch.view_as_complex(rr.reshape(-1, 2))
    out, at = [], 0
    for c in recv_counts:
        out.append(flat[at:at + c])
        at += c
    return out


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from marlin_amd.api import pencil_factors, pencil_layout
        from oracle import marlin_oracle as mo
        Py, Pz = pencil_factors(world, n)
        lay = [pencil_layout(world, r, n) for r in range(world)]
        me = lay[rank]
        px, pz = rank % Py, rank // Py
        torch.manual_seed(21)
        g = torch.rand(n, dtype=torch.float64)
        sl = tuple(slice(b, b + s) for b, s in zip(me["real_begin"], me["real_shape"]))
        blk = g[sl].contiguous()
        kxl, kyl = me["recip_shape"][0], me["recip_shape"][1]
        # ---- forward: rfft x -> stage 1 -> fft y -> stage 2 -> fft z
        ax = torch.fft.rfft(blk, dim=0)
        group1 = [pz * Py + q_ for q_ in range(Py)]
        send = [torch.zeros(0, dtype=torch.complex128)] * world
        for p in group1:
            kb, ks = lay[p]["recip_begin"][0], lay[p]["recip_shape"][0]
            send[p] = ax[kb:kb + ks].contiguous()
        got = _exchange(send, me["stage1_send"], me["stage1_recv"])
        a1 = torch.empty((kxl, n[1], me["real_shape"][2]), dtype=torch.complex128)
        for p in group1:
            yb, ys = lay[p]["real_begin"][1], lay[p]["real_shape"][1]
            a1[:, yb:yb + ys, :] = got[p].reshape(kxl, ys, me["real_shape"][2])
        a1 = torch.fft.fft(a1, dim=1)
        group2 = [q_ * Py + px for q_ in range(Pz)]
        send = [torch.zeros(0, dtype=torch.complex128)] * world
        for p in group2:
            kb, ks = lay[p]["recip_begin"][1], lay[p]["recip_shape"][1]
            send[p] = a1[:, kb:kb + ks, :].contiguous()
        got = _exchange(send, me["stage2_send"], me["stage2_recv"])
        a2 = torch.empty((kxl, kyl, n[2]), dtype=torch.complex128)
        for p in group2:
            zb, zs = lay[p]["real_begin"][2], lay[p]["real_shape"][2]
            a2[:, :, zb:zb + zs] = got[p].reshape(kxl, kyl, zs)
        spec = torch.fft.fft(a2, dim=2)
        full = torch.fft.fftn(torch.fft.rfft(g, dim=0), dim=(1, 2))
        ks = tuple(slice(b, b + s) for b, s in zip(me["recip_begin"], me["recip_shape"]))
        e_fwd = (spec - full[ks]).abs().max().item() / full.abs().max().item()
        # ---- inverse: ifft z -> stage 2 (sizes swapped) -> ifft y -> stage 1 (sizes swapped) -> irfft x
        b2 = torch.fft.ifft(spec, dim=2)
        send = [torch.zeros(0, dtype=torch.complex128)] * world
        for p in group2:
            zb, zs = lay[p]["real_begin"][2], lay[p]["real_shape"][2]
            send[p] = b2[:, :, zb:zb + zs].contiguous()
        got = _exchange(send, me["stage2_recv"], me["stage2_send"])
        b1 = torch.empty((kxl, n[1], me["real_shape"][2]), dtype=torch.complex128)
        for p in group2:
            kb, ks_ = lay[p]["recip_begin"][1], lay[p]["recip_shape"][1]
            b1[:, kb:kb + ks_, :] = got[p].reshape(kxl, ks_, me["real_shape"][2])
        b1 = torch.fft.ifft(b1, dim=1)
        send = [torch.zeros(0, dtype=torch.complex128)] * world
        for p in group1:
            yb, ys = lay[p]["real_begin"][1], lay[p]["real_shape"][1]
            send[p] = b1[:, yb:yb + ys, :].contiguous()
        got = _exchange(send, me["stage1_recv"], me["stage1_send"])
        half = torch.empty((n[0] // 2 + 1, me["real_shape"][1], me["real_shape"][2]), dtype=torch.complex128)
        for p in group1:
            kb, ks_ = lay[p]["recip_begin"][0], lay[p]["recip_shape"][0]
            half[kb:kb + ks_] = got[p].reshape(ks_, me["real_shape"][1], me["real_shape"][2])
        back = torch.fft.irfft(half, n=n[0], dim=0)
        e_inv = (back - blk).abs().max().item()
        # the library's layout == the oracle's restatement of partitionPencils
        d = mo.PencilDomain(n, [1.0, 1.0, 1.0], world)
        same = (me["real_begin"] == [s.start for s in d.real_slices(rank)] and me["recip_begin"] == [s.start for s in d.recip_slices(rank)]
                and me["recip_shape"] == [s.stop - s.start for s in d.recip_slices(rank)])
        q.put((rank, e_fwd, e_inv, same))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(4, [16, 12, 10]), (4, [9, 8, 7]), (6, [12, 10, 9])])
def test_pencil_exchange_pattern_over_gloo(world, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    for rank, e_fwd, e_inv, same in res:
        assert same and e_fwd <= 2e-15 * max(n) and e_inv <= 1e-14, (rank, e_fwd, e_inv, same)
