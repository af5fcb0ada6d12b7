#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz from the reference's own gold files.

Run in the build container only (needs /root/reference and /opt/conda/bin/h5dump;
neither exists on the GPU box).  The .npz files are DATA: the datasets of the
reference's committed HDF5 gold files, converted losslessly (raw little-endian
IEEE f64 dumped by `h5dump -b LE`), plus the CSV gold tables of the solver tests.

Sources (relative to /root/reference):
  test/tests/cahnhilliard/gold/cahnhilliard.h5           spec test/tests/cahnhilliard/tests:46-57  (abs_tol 1e-13)
  test/tests/cahnhilliard/gold/cahnhilliard.rank0001.h5  spec test/tests/cahnhilliard/tests:58-70  (2-rank FFT_SLAB, rank 1)
  test/tests/cahnhilliard/gold/cahnhilliard.xmf          spec test/tests/cahnhilliard/tests:35-45  (XMLDiff of the XDMF description; stored as a JSON tree)
  test/tests/mechanics/gold/mech3d.h5, mech.h5           spec test/tests/mechanics/tests:2-21      (abs_tol 1e-10)
  test/tests/tensor_compute/gold/rotating_grain_secant.h5 spec test/tests/tensor_compute/tests:90-100 (abs_tol 1e-10)
  test/tests/cahnhilliard/gold/map_to_aux_3d.e            spec test/tests/cahnhilliard/tests:13-22 (3-D 5^3 Cahn-Hilliard; Exodus)
  test/tests/cahnhilliard/gold/sharp.e, houli.e           spec test/tests/cahnhilliard/tests:121-143 (explicit Euler + DeAliasingTensor; Exodus)
  test/tests/kks/gold/KKS_no_flux_bc.h5, _out.csv         spec test/tests/kks/tests:13-31 (2-variable KKS, smooth boundary method, abs_tol 1e-10)
  test/tests/typed_tensors/gold/gradient.h5               spec test/tests/typed_tensors/tests (GradientTensor, 20x10x5, NODE mode)
  test/tests/solvers/gold/*.csv                          spec test/tests/solvers/tests
  test/tests/tensor_compute/gold/backandforth_out.csv, test/tests/gradient/gold/gradient_out.csv
"""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

REF = "/root/reference"
H5DUMP = "/opt/conda/bin/h5dump"
OUT = os.path.dirname(os.path.abspath(__file__))


def xml_tree(path):
    """an XML file as nested [tag, attributes, text, children] lists (what XMLDiff compares: structure, attributes, text)"""
    import xml.etree.ElementTree as ET

    def walk(e):
        return [e.tag, dict(sorted(e.attrib.items())), (e.text or "").strip(), [walk(c) for c in e]]
    return walk(ET.parse(path).getroot())


def convert_xmf(rel, out):
    import json
    with open(os.path.join(OUT, out), "w") as f:
        json.dump(xml_tree(os.path.join(REF, rel)), f, indent=0)
    print("wrote", out)


def h5_names(path):
    txt = subprocess.check_output([H5DUMP, "-n", path], text=True)
    return re.findall(r"dataset\s+/(\S+)", txt)


def h5_shape(path, name):
    txt = subprocess.check_output([H5DUMP, "-H", "-d", "/" + name, path], text=True)
    m = re.search(r"SIMPLE \{ \(([^)]*)\)", txt)
    assert "H5T_IEEE_F64LE" in txt, txt
    return tuple(int(s) for s in m.group(1).split(","))


def h5_read(path, name):
    shape = h5_shape(path, name)
    with tempfile.NamedTemporaryFile(suffix=".bin") as tmp:
        subprocess.check_call(
            [H5DUMP, "-d", "/" + name, "-b", "LE", "-o", tmp.name, path],
            stdout=subprocess.DEVNULL,
        )
        a = np.fromfile(tmp.name, dtype="<f8")
    assert a.size == int(np.prod(shape)), (name, a.size, shape)
    return a.reshape(shape)


def convert_h5(rel, out_name):
    path = os.path.join(REF, rel)
    data = {n: h5_read(path, n) for n in h5_names(path)}
    np.savez_compressed(os.path.join(OUT, out_name), **data)
    print(f"{out_name}: {len(data)} datasets from {rel}")


def convert_exodus(rel, out_name, n, length, frames, dim=2):
    """nodal / elemental variables of a MOOSE Exodus file (netCDF classic, read with scipy) back onto the n^dim tensor grid:
    ProjectTensorAux (src/auxkernels/ProjectTensorAux.C:36-71) puts cell (i, j[, k]) on node (i, j[, k]) (periodic wrap at
    i = n) and on element (i, j[, k])"""
    from scipy.io import netcdf_file
    f = netcdf_file(os.path.join(REF, rel), "r", mmap=False)
    xyz = [f.variables["coord" + a][:] for a in "xyz"[:dim]]
    dx = length / n
    ni = tuple(np.rint(a / dx).astype(int) for a in xyz)
    conn = f.variables["connect1"][:] - 1
    ei = tuple(np.floor(a[conn].mean(1) / dx).astype(int) for a in xyz)
    nod, el = f.variables["vals_nod_var1"][:], f.variables["vals_elem_var1eb1"][:]
    data = {"time": np.array(f.variables["time_whole"][:])[list(frames)]}
    inner = (slice(0, n),) * dim
    for k in frames:
        c = np.zeros((n + 1,) * dim)
        c[ni] = nod[k]
        m = np.zeros((n,) * dim)
        m[ei] = el[k]
        assert np.array_equal(c[(n,) + inner[1:]], c[(0,) + inner[1:]])
        data[f"c.{k}"], data[f"mu.{k}"] = c[inner].copy(), m
    np.savez_compressed(os.path.join(OUT, out_name), **data)
    print(f"{out_name}: frames {list(frames)} of {rel}")


def convert_csv(rels, out_name):
    data = {}
    for rel in rels:
        path = os.path.join(REF, rel)
        with open(path) as f:
            header = f.readline().strip().split(",")
        a = np.loadtxt(path, delimiter=",", skiprows=1, ndmin=2)
        key = os.path.splitext(os.path.basename(rel))[0]
        data[key] = a
        data[key + "__columns"] = np.array(header)
    np.savez_compressed(os.path.join(OUT, out_name), **data)
    print(f"{out_name}: {len(rels)} csv tables")


def main():
    convert_h5("test/tests/cahnhilliard/gold/cahnhilliard.h5", "cahnhilliard_gold.npz")
    convert_xmf("test/tests/cahnhilliard/gold/cahnhilliard.xmf", "cahnhilliard_xmf_gold.json")   # spec tests:35-45 (XMLDiff)
    convert_h5("test/tests/cahnhilliard/gold/cahnhilliard.rank0001.h5", "cahnhilliard_rank0001_gold.npz")
    convert_h5("test/tests/mechanics/gold/mech3d.h5", "mech3d_gold.npz")
    convert_h5("test/tests/mechanics/gold/mech.h5", "mech2d_gold.npz")
    convert_h5("test/tests/tensor_compute/gold/rotating_grain_secant.h5", "rotating_grain_secant_gold.npz")
    convert_h5("test/tests/typed_tensors/gold/gradient.h5", "typed_gradient_gold.npz")
    convert_h5("test/tests/kks/gold/KKS_no_flux_bc.h5", "kks_no_flux_bc_gold.npz")
    convert_h5("test/tests/tensor_compute/gold/smooth_rectangle.h5", "smooth_rectangle_gold.npz")
    # cahnhilliard.i with Domain/dim=3 nx=ny=nz=5 zmax=3 (tests:13-22): the only 3-D Cahn-Hilliard gold data of the reference
    convert_exodus("test/tests/cahnhilliard/gold/map_to_aux_3d.e", "cahnhilliard_3d_gold.npz", 5, 3.0, range(11), dim=3)
    for m in ("sharp", "houli"):      # cahnhilliard_explicit_smooth.i with DeAliasingTensor method = SHARP / HOULI (Exodiff)
        convert_exodus(f"test/tests/cahnhilliard/gold/{m}.e", f"cahnhilliard_explicit_{m}_gold.npz", 50, 3.0, (0, 1, 2, 5, 10, 20))
    sol = sorted(
        os.path.join("test/tests/solvers/gold", f)
        for f in os.listdir(os.path.join(REF, "test/tests/solvers/gold"))
        if f.endswith(".csv")
    )
    convert_csv(sol, "solvers_gold.npz")
    convert_csv(
        ["test/tests/tensor_compute/gold/backandforth_out.csv", "test/tests/gradient/gold/gradient_out.csv",
         "test/tests/gradient/gold/gradient_square_out.csv", "test/tests/kks/gold/KKS_no_flux_bc_out.csv",
         "test/tests/postprocessors/gold/interface_velocity_out.csv", "test/tests/histogram/gold/test_out_hist_0001.csv"],
        "fft_gold.npz",
    )


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    main()
