#!/usr/bin/env python3
"""Generate tests/golden/*.json from the reference's own test data.

Run in the build container only (needs /root/reference and /opt/conda/bin/h5dump):

    python tests/golden/make_fixtures.py

Outputs (data only -- inputs and expected outputs, no reference source text):
  random_polish_qp.json   the JLD2 fixture [REF test/problem_data/random_polish_qp.jld2]
                          used by [REF test/polishing.jl:69-93]: P (dense 30x30, both
                          triangles), q, A (50x30, 743 nnz), l, u and the Mosek-accurate
                          x_test, y_test, obj_test.
  known_answers.json      the literal problems and expected values of test/basic.jl,
                          polishing.jl, non_convex.jl, dual_infeasibility.jl,
                          primal_infeasibility.jl, MOI_wrapper.jl:283-332.

h5py is not installed; dense datasets are read with `h5dump`, and the two
SparseMatrixCSC compounds (whose fields are object references = file addresses
relative to the 512-byte user block) are followed with a small parser of HDF5
version-2 object headers.
"""
import json
import os
import re
import struct
import subprocess

REF = "/root/reference"
JLD = os.path.join(REF, "test/problem_data/random_polish_qp.jld2")
H5DUMP = "/opt/conda/bin/h5dump"
HERE = os.path.dirname(os.path.abspath(__file__))
USERBLOCK = 512


def h5dump_dense(name):
    out = subprocess.check_output([H5DUMP, "-m", "%.17g", "-d", "/" + name, JLD], text=True)
    body = out[out.index("DATA {") + 6:]
    body = body[: body.index("}")]
    vals = []
    for line in body.splitlines():
        line = re.sub(r"^\s*\(\d+\):", "", line)
        for tok in line.replace(",", " ").split():
            vals.append(float(tok))
    return vals


def h5dump_compound(name):
    out = subprocess.check_output([H5DUMP, "-d", "/" + name, JLD], text=True)
    body = out[out.index("(0): {") + 6:]
    nums = re.findall(r"(?:DATASET\s+)?(\d+)", body[: body.index("}")])
    m, n, a, b, c = (int(v) for v in nums[:5])
    return m, n, a, b, c


def read_object(buf, addr):
    """Parse a version-2 object header at file address `addr` (already including the
    user block) and return (dims, class, size, raw bytes) of its contiguous/compact data."""
    assert buf[addr:addr + 4] == b"OHDR", "not a v2 object header"
    version, flags = buf[addr + 4], buf[addr + 5]
    assert version == 2
    p = addr + 6
    if flags & 0x20:
        p += 16  # access/modification/change/birth times
    if flags & 0x10:
        p += 4   # max compact / min dense attributes
    size_bytes = 1 << (flags & 0x3)
    chunk_size = int.from_bytes(buf[p:p + size_bytes], "little")
    p += size_bytes
    end = p + chunk_size
    dims, tclass, tsize, data = None, None, None, None
    while p + 4 <= end:
        mtype = buf[p]
        msize = struct.unpack_from("<H", buf, p + 1)[0]
        p += 4
        if flags & 0x04:
            p += 2  # creation order
        body = buf[p:p + msize]
        if mtype == 0x01:  # dataspace
            ver, rank, dflags = body[0], body[1], body[2]
            off = 4 if ver == 2 else 8
            dims = [struct.unpack_from("<Q", body, off + 8 * k)[0] for k in range(rank)]
        elif mtype == 0x03:  # datatype
            tclass = body[0] & 0x0F
            tsize = struct.unpack_from("<I", body, 4)[0]
        elif mtype == 0x08:  # layout
            ver, lclass = body[0], body[1]
            assert ver in (3, 4)
            if lclass == 1:  # contiguous
                daddr, dsize = struct.unpack_from("<QQ", body, 2)
                data = buf[daddr + USERBLOCK: daddr + USERBLOCK + dsize]
            elif lclass == 0:  # compact
                dsize = struct.unpack_from("<H", body, 2)[0]
                data = body[4:4 + dsize]
            else:
                raise RuntimeError("chunked layout not expected")
        p += msize
    return dims, tclass, tsize, data


def read_vector(buf, ref):
    dims, tclass, tsize, data = read_object(buf, ref + USERBLOCK)
    count = 1
    for d in dims:
        count *= d
    assert tsize == 8
    fmt = "<%d%s" % (count, "q" if tclass == 0 else "d")
    return list(struct.unpack(fmt, data[: 8 * count]))


def read_sparse(buf, name):
    m, n, a, b, c = h5dump_compound(name)
    colptr = read_vector(buf, a)
    rowval = read_vector(buf, b)
    nzval = read_vector(buf, c)
    assert len(colptr) == n + 1 and len(rowval) == len(nzval) == colptr[-1] - 1
    # to 0-based, as the C ABI wants it [REF src/types.jl:39-43]
    return {"m": m, "n": n, "p": [v - 1 for v in colptr], "i": [v - 1 for v in rowval], "x": nzval}


def main():
    buf = open(JLD, "rb").read()
    fx = {
        "source": "test/problem_data/random_polish_qp.jld2 (osqp/OSQP.jl v0.8.1)",
        "P": read_sparse(buf, "P"),
        "A": read_sparse(buf, "A"),
    }
    for name in ("q", "l", "u", "x_test", "y_test"):
        fx[name] = h5dump_dense(name)
    fx["obj_test"] = h5dump_dense("obj_test")[0]
    with open(os.path.join(HERE, "random_polish_qp.json"), "w") as f:
        json.dump(fx, f)
    print("random_polish_qp.json: nnz(P) =", len(fx["P"]["x"]), "nnz(A) =", len(fx["A"]["x"]))

    inf = "inf"
    basic = {
        "P": [[11.0, 0.0], [0.0, 0.0]], "q": [3.0, 4.0],
        "A": [[-1, 0], [0, -1], [-1, -3], [2, 5], [3, 4]],
        "l": ["-inf"] * 5, "u": [0.0, 0.0, -15.0, 100.0, 80.0],
        "options": {"verbose": False, "eps_abs": 1e-9, "eps_rel": 1e-9, "check_termination": 1, "polish": False,
                    "max_iter": 4000, "rho": 0.1, "adaptive_rho": False, "warm_start": True},
    }
    ka = {
        "_comment": "known answers restated from the reference's tests (SURVEY.md Appendix B); file:line in 'ref'",
        "basic": basic,
        "G1": {"ref": "test/basic.jl:43-49", "x": [0.0, 5.0], "y": [1.666666666666, 0.0, 1.3333333, 0.0, 0.0], "obj": 20.0, "tol": 1e-5},
        "G2": {"ref": "test/basic.jl:66-75", "q": [10.0, 20.0], "x": [0.0, 5.0], "y": [3.33333333, 0.0, 6.66666666, 0.0, 0.0], "obj": 100.0, "tol": 1e-5},
        "G3": {"ref": "test/basic.jl:92-101", "l": [-100.0] * 5, "x": [0.0, 5.0], "y": [1.666666666666, 0.0, 1.3333333, 0.0, 0.0], "obj": 20.0, "tol": 1e-5},
        "G4": {"ref": "test/basic.jl:118-131", "u": [1000.0] * 5, "x": [-0.151515152, -333.282828], "y": [0.0, 0.0, 1.33333333, 0.0, 0.0], "obj": -1333.459595961, "tol": 1e-5},
        "G5": {"ref": "test/basic.jl:148-151", "max_iter": 80, "status": "Max_iter_reached"},
        "G6": {"ref": "test/basic.jl:168-171", "check_termination": 0, "iter": 4000},
        "G7": {"ref": "test/basic.jl:189-207", "rho_setup": 0.7, "rho_update": 0.1},
        "G8": {"ref": "test/basic.jl:228-239", "eps_abs": 1e-20, "eps_rel": 1e-20, "time_limit": 1e-6, "max_iter": 1000000, "check_termination": 0, "status": "Time_limit_reached"},
        "G9": {"ref": "test/polishing.jl:17-37", "options": {"verbose": False, "polish": True, "eps_abs": 1e-3, "eps_rel": 1e-3, "max_iter": 5000},
               "x": [9.90341e-11, 5.0], "y": [1.66667, 0.0, 1.33333, 1.20431e-14, 1.49741e-14], "obj": 20.0, "status_polish": 1, "tol": 1e-3},
        "G11": {"ref": "test/non_convex.jl:6-21", "P": [[2.0, 5.0], [5.0, 1.0]], "sigma": 1e-6, "setup_fails": True},
        "G12": {"ref": "test/non_convex.jl:27-40", "P": [[2.0, 5.0], [5.0, 1.0]], "sigma": 5.0, "status": "Non_convex", "obj_nan": True},
        "dual_inf_options": {"verbose": False, "eps_abs": 1e-5, "eps_rel": 1e-5, "eps_prim_inf": 1e-15, "check_termination": 1},
        "G13": {"ref": "test/dual_infeasibility.jl:16-27", "P": [[0.0, 0.0], [0.0, 0.0]], "q": [2.0, -1.0], "A": [[1, 0], [0, 1]],
                "l": [0.0, 0.0], "u": [inf, inf], "status": "Dual_infeasible"},
        "G14": {"ref": "test/dual_infeasibility.jl:31-42", "P": [[4.0, 0.0], [0.0, 0.0]], "q": [0.0, 2.0], "A": [[1.0, 1.0], [-1.0, 1.0]],
                "l": ["-inf", "-inf"], "u": [2.0, 3.0], "status": "Dual_infeasible"},
        "G15": {"ref": "test/dual_infeasibility.jl:46-61", "P": [[0.0, 0.0], [0.0, 0.0]], "q": [-1.0, -1.0],
                "A": [[1.0, -1.0], [-1.0, 1.0], [1.0, 0.0], [0.0, 1.0]], "l": [1.0, 1.0, 0.0, 0.0], "u": [inf] * 4,
                "warm_x": [50.0, 30.0], "warm_y": [-2.0, -2.0, -2.0, -2.0], "status": "Dual_infeasible"},
        "prim_inf_options": {"verbose": False, "eps_abs": 1e-5, "eps_rel": 1e-5, "eps_dual_inf": 1e-18, "scaling": 1},
        "G16": {"ref": "test/primal_infeasibility.jl:44-58", "P": [[0.0, 0.0], [0.0, 0.0]], "q": [-1.0, -1.0],
                "A": [[1.0, -1.0], [-1.0, 1.0], [1.0, 0.0], [0.0, 1.0]], "l": [1.0, 1.0, 0.0, 0.0], "u": [inf] * 4,
                "status": "Primal_infeasible"},
        "G18": {"ref": "test/MOI_wrapper.jl:283-332", "P": [[0.0, 0.0], [0.0, 0.0]], "q": [-1.0, 0.0],
                "A": [[1.0, 1.0], [1.0, 0.0], [0.0, 1.0]], "l": ["-inf", 0.0, 0.0], "u": [1.0, inf, inf],
                "options": {"verbose": False, "eps_abs": 1e-8, "eps_rel": 1e-16, "max_iter": 10000, "adaptive_rho_interval": 25},
                "x": [1.0, 0.0], "obj": -1.0, "moi_duals": [-1.0, 0.0, 1.0], "tol": 1e-4},
    }
    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump(ka, f, indent=1)
    print("known_answers.json written")


if __name__ == "__main__":
    main()
