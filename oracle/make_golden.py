#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (oracle).  Generate tests/golden/*.json by RUNNING THE REAL
REFERENCE (stfc/dl_esm_inf, serial build compiled into oracle/_ref by
`make -C oracle ref`) through the dump driver oracle/ref_drivers/ref_dump.f90 and
through the reference's own example / device-io test programs.

Only runs in the build container (needs /root/reference to have been compiled
into oracle/_ref); the JSON it writes is committed, the binaries are not.

    python oracle/make_golden.py
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "_ref")
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def run(exe, *args, env_extra=None):
    env = dict(os.environ)
    env.pop("DL_ESM_ALIGNMENT", None)
    if env_extra:
        env.update(env_extra)
    p = subprocess.run([os.path.join(REF, exe), *map(str, args)], env=env,
                       capture_output=True, text=True, timeout=120)
    return p.returncode, p.stdout, p.stderr


def g_lines(stdout):
    out = {}
    for line in stdout.splitlines():
        if line.startswith("G: "):
            toks = line[3:].split()
            out.setdefault(toks[0], []).append(toks[1:])
    return out


def ints(toks):
    return [int(t) for t in toks]


def golden_decomp():
    cases = [(4, 10, 1), (10, 4, 2), (4, 10, 2), (10, 10, 4), (10, 10, 6),
             (4, 10, 4), (4, 10, 6), (16, 32, 8), (32, 16, 8), (10, 10, 3),
             (7, 5, 2), (12, 9, 5), (100, 37, 7), (37, 100, 7), (256, 256, 1),
             (8192, 8192, 1), (8192, 16384, 2), (16384, 16384, 4),
             (16384, 32768, 8), (32768, 16384, 8), (16384, 16384, 1),
             (13, 13, 9), (64, 64, 16), (10, 10, 2), (9, 10, 2), (10, 9, 2)]
    res = []
    for nx, ny, nd in cases:
        rc, so, se = run("ref_dump.exe", "decomp", nx, ny, nd)
        assert rc == 0, (nx, ny, nd, se)
        g = g_lines(so)
        d = ints(g["decomp"][0])
        subs = [ints(s)[1:] for s in g["sub"]]
        res.append({
            "domainx": nx, "domainy": ny, "ndomains": nd,
            "global_nx": d[0], "global_ny": d[1], "nx": d[2], "ny": d[3],
            "ndom_out": d[4], "max_width": d[5], "max_height": d[6],
            # per subdomain: global xstart,xstop,ystart,ystop,nx,ny ; internal xstart,xstop,ystart,ystop,nx,ny
            "subdomains": subs,
        })
    return {"_provenance": "reference go_decompose (parallel_mod.f90:70-332) run via "
                           "oracle/_ref/ref_dump.exe decomp", "cases": res}


def golden_bounds():
    sizes = [(10, 4), (4, 10), (10, 10), (256, 256), (5, 5)]
    # (offset, bcx, bcy): GO_OFFSET_NE=3, GO_OFFSET_SW=0 ; GO_BC_PERIODIC=0, EXTERNAL=1, NONE=2
    combos = [(3, 1, 1), (0, 0, 0), (0, 1, 1), (3, 0, 0), (3, 2, 2), (0, 0, 1), (0, 1, 0)]
    aligns = [None, 1, 8, 64]
    res = []
    for nx, ny in sizes:
        for off, bcx, bcy in combos:
            for pt in range(5):
                for al in aligns:
                    env = {"DL_ESM_ALIGNMENT": str(al)} if al else None
                    rc, so, se = run("ref_dump.exe", "bounds", nx, ny, off, bcx, bcy, pt,
                                     env_extra=env)
                    g = g_lines(so)
                    rec = {"nx": nx, "ny": ny, "offset": off, "bcx": bcx, "bcy": bcy,
                           "ptype": pt, "alignment": al}
                    if "field" in g:
                        f = ints(g["field"][0])
                        rec.update({
                            "abort": False,
                            "grid": ints(g["grid"][0]),       # grid%nx, grid%ny, global_nx, global_ny
                            "defined_on": f[0],
                            "internal": f[1:7],               # xstart,xstop,ystart,ystop,nx,ny
                            "whole": f[7:13],
                            "num_halos": f[13],
                            "shape": ints(g["shape"][0]),
                            "halos": [ints(h)[1:] for h in g.get("halo", [])],
                        })
                        # num_halos is left uninitialised by the reference for some
                        # NE fields (field_mod.f90:872-895 never sets it): not a contract.
                        if off == 3 and pt in (1, 2, 3):
                            rec["num_halos"] = None
                    else:
                        rec.update({"abort": True,
                                    "grid": ints(g["grid"][0]) if "grid" in g else None,
                                    "message": se.strip().splitlines()[0].strip() if se.strip() else ""})
                    res.append(rec)
    return {"_provenance": "reference grid_init (grid_mod.f90:349-385) + set_field_bounds "
                           "(field_mod.f90:563-1122) run via oracle/_ref/ref_dump.exe bounds",
            "cases": res}


def golden_model():
    res = {"_provenance": "reference example (example/model.f90) compiled unmodified -> "
                          "ref_example.exe, plus ref_dump.exe model/gather (config-1 plumbing)"}
    rc, so, se = run("ref_example.exe")
    assert rc == 0
    res["example_4x10"] = {k: float(v) for k, v in
                           (l.replace(" checksum =", "").split() for l in so.splitlines()
                            if "checksum =" in l)}
    res["model"] = []
    for nx, ny, fill in [(4, 10, 1.0), (256, 256, 1.0), (64, 48, 2.5)]:
        rc, so, se = run("ref_dump.exe", "model", nx, ny, fill)
        assert rc == 0
        g = g_lines(so)
        res["model"].append({
            "nx": nx, "ny": ny, "fill": fill,
            "grid": ints(g["grid"][0]), "internal": ints(g["internal"][0]),
            "checksum": float(g["checksum"][0][0]),
            "xt": [float(x) for x in g["xt"][0]], "yt": [float(x) for x in g["yt"][0]],
        })
    res["gather"] = []
    for nx, ny in [(10, 10), (4, 10), (33, 17)]:
        rc, so, se = run("ref_dump.exe", "gather", nx, ny)
        assert rc == 0
        g = g_lines(so)
        res["gather"].append({
            "nx": nx, "ny": ny,
            "corner": [float(x) for x in g["corner"][0]],
            "checksum": float(g["checksum"][0][0]),
            "gather_shape": ints(g["gather_shape"][0]),
            "gather_mismatch": int(g["gather_mismatch"][0][0]),
        })
    return res


def golden_tmask():
    """grid%tmask after grid_init(tmask = a -1/0/1 pattern): copy-in + boundary fill"""
    res = {"_provenance": "reference grid_init (grid_mod.f90:394-432) with tmask(i,j) = mod(7i+13j,3)-1 run via "
                          "oracle/_ref/ref_dump.exe tmask; rows are grid%tmask(1:nx, j)", "cases": []}
    for nx, ny, al in [(10, 4, None), (10, 4, 8), (7, 9, 64), (16, 5, 4), (4, 10, 1)]:
        env = {"DL_ESM_ALIGNMENT": str(al)} if al else None
        rc, so, se = run("ref_dump.exe", "tmask", nx, ny, env_extra=env)
        assert rc == 0, se
        g = g_lines(so)
        rows = sorted((ints(r) for r in g["tmaskrow"]), key=lambda r: r[0])
        res["cases"].append({"nx": nx, "ny": ny, "alignment": al, "grid": ints(g["grid"][0]),
                             "tmask": [r[1:] for r in rows]})
    return res


def golden_device_io():
    res = {"_provenance": "reference tests/device_computation/test_device_io.f90 compiled "
                          "unmodified -> ref_device_io.exe; 'Resulting array' rows as printed",
           "runs": []}
    for al in [None, 2, 8]:
        env = {"DL_ESM_ALIGNMENT": str(al)} if al else None
        rc, so, se = run("ref_device_io.exe", env_extra=env)
        lines = so.splitlines()
        i = next(k for k, l in enumerate(lines) if "Resulting array" in l)
        rows = [[float(x) for x in l.split()] for l in lines[i + 1:i + 9]]
        ops = [l.split() for l in lines if "operation" in l]
        res["runs"].append({"alignment": al, "rc": rc, "rows": rows,
                            "ops": ops, "passed": "Test passed" in so})
    return res


def main():
    if not os.path.exists(os.path.join(REF, "ref_dump.exe")):
        sys.exit("oracle/_ref not built: run `make -C oracle ref` in the build container")
    os.makedirs(OUT, exist_ok=True)
    for name, fn in [("ref_decomp", golden_decomp), ("ref_bounds", golden_bounds),
                     ("ref_model", golden_model), ("ref_device_io", golden_device_io),
                     ("ref_tmask", golden_tmask)]:
        data = fn()
        with open(os.path.join(OUT, name + ".json"), "w") as f:
            json.dump(data, f, indent=None, separators=(",", ":"))
            f.write("\n")
        print("wrote", name)


if __name__ == "__main__":
    main()
