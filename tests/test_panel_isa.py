"""Build-time census of the panel kernels' ISA (gpirt_amd/csrc/panel.hip, gfx950).

The persistent panel kernel hands bytes between work-groups WITHOUT agent-scope fences: it is correct only while every
handed-off byte is stored and loaded `sc1` (flagsync.h).  Nothing in the language guards that -- a plain load on a
shared path would give a silently wrong factor, not an abort (round-2 advisor finding).  So the ISA of every kernel in
the file is counted -- global loads / stores with and without sc1, fences (buffer_wbl2 / buffer_inv), scratch, registers
-- and compared with a committed census (tests/golden/panel_isa_census.json).  A change in the number of PLAIN accesses
fails this test: the author then checks that the new access is to rows only its own work-group reads (stg_plain /
ldg_off in panel.hip), and refreshes the census with  python tests/test_panel_isa.py --update.
sc1 counts may grow freely (more shared traffic is safe); fences and scratch must stay zero."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CENSUS = os.path.join(ROOT, "tests", "golden", "panel_isa_census.json")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-function", "-Wno-unused-value"]
NAMES = {"panel_ll_kernel": "panel_ll_kernel"}


def census():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "panel.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "--cuda-device-only", "-S",
                               os.path.join(ROOT, "gpirt_amd", "csrc", "panel.hip"), "-o", out])
        txt = open(out).read()
    res = {}
    for m in re.finditer(r"^(_ZN5gpirt\S+):\s*;\s*@\S+\n(.*?)\n\s*\.end_amdhsa_kernel", txt, re.S | re.M):
        sym, body = m.group(1), m.group(2)
        name = next((v for k, v in NAMES.items() if k in sym), None)
        if name is None:
            continue
        ins = [ln.split(";")[0].strip() for ln in body.splitlines()]
        ld = [i for i in ins if i.startswith(("global_load", "flat_load", "buffer_load"))]
        st = [i for i in ins if i.startswith(("global_store", "flat_store", "buffer_store"))]
        at = [i for i in ins if i.startswith(("global_atomic", "flat_atomic"))]
        res[name] = {
            "loads_sc1": sum("sc1" in i for i in ld), "loads_plain": sum("sc1" not in i for i in ld),
            "stores_sc1": sum("sc1" in i for i in st), "stores_plain": sum("sc1" not in i for i in st),
            "atomics": len(at),
            "flat_or_buffer_accesses": sum(i.startswith(("flat_", "buffer_load", "buffer_store")) for i in ld + st),
            "buffer_wbl2": sum(i.startswith("buffer_wbl2") for i in ins), "buffer_inv": sum(i.startswith("buffer_inv") for i in ins),
            "scratch": sum(i.startswith("scratch_") for i in ins),
            "mfma": sum(i.startswith("v_mfma_f64") for i in ins),
            "next_free_vgpr": int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body[body.rfind(".amdhsa_kernel") if ".amdhsa_kernel" in body else 0:] or body).group(1))
            if re.search(r"\.amdhsa_next_free_vgpr (\d+)", body) else None,
        }
    return res


def test_panel_isa_census_unchanged():
    got = census()
    want = json.load(open(CENSUS))
    assert set(got) == set(want), (sorted(got), sorted(want))
    for k in want:
        g, w = got[k], want[k]
        # the fence-free form of the hand-offs: not one write-back or invalidate of the XCD's L2 in the kernel
        assert g["buffer_wbl2"] == 0 and g["buffer_inv"] == 0, (k, g)
        assert g["scratch"] == 0, (k, g)                                        # nothing spilled to memory
        assert g["flat_or_buffer_accesses"] == 0, (k, g)                        # sc1 hand-offs must be global_ (never flat_)
        assert g["loads_plain"] == w["loads_plain"] and g["stores_plain"] == w["stores_plain"], (
            f"{k}: the number of PLAIN global accesses changed ({w['loads_plain']} loads / {w['stores_plain']} stores -> "
            f"{g['loads_plain']} / {g['stores_plain']}): check that the new access is to own rows only, then refresh the census")
        assert g["loads_sc1"] >= 1 and g["mfma"] >= 1, (k, g)


if __name__ == "__main__":
    c = census()
    if "--update" in sys.argv:
        json.dump(c, open(CENSUS, "w"), indent=1, sort_keys=True)
    print(json.dumps(c, indent=1, sort_keys=True))
