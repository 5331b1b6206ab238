"""Build-time check that the register-resident kernels really are register-resident (gfx950 ISA of rng_ess.hip /
rs_predict.hip): a kernel whose per-lane arrays slip into scratch memory still gives the right numbers -- round 6 lost 5 %
of the headline for an afternoon that way (capturing lambdas around the slice kernel's arrays: 164 registers + 432 bytes of
scratch per lane instead of 237 registers, 330-450 instead of 140-205 us).  The kernels named here must compile with no
scratch at all; the others of the two files are listed with what they use so that a change shows in the log."""
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-function", "-Wno-unused-value"]
# (mangled-name fragment, why it matters)
NO_SCRATCH = {
    "rng_ess.hip": ["14ess_kernel_regILi8ELi256ELb1ELb0E", "14ess_kernel_regILi16ELi512ELb1ELb0E", "19rs3_products_kernel"],
    "rs_predict.hip": ["20rs3p_products_kernel", "23rs3p_products_lr_kernel", "18rs3p_decide_kernel", "16rs_verify_kernel", "20rs_verify_reg_kernelILi16ELi512E", "20rs_verify_reg_kernelILi8ELi256E"],
    "rs_lr.hip": ["18rs_lr_apply_kernel", "14lr_coef_kernel"],
}


def _kernels(src):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "--cuda-device-only", "-S", os.path.join(ROOT, "gpirt_amd", "csrc", src), "-o", out])
        txt = open(out).read()
    res = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", txt, re.S):
        body = m.group(2)
        priv = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body)
        vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body)
        res[m.group(1)] = (int(priv.group(1)) if priv else 0, int(vg.group(1)) if vg else None)
    return res


def test_register_resident_kernels_use_no_scratch():
    for src, wanted in NO_SCRATCH.items():
        ks = _kernels(src)
        for frag in wanted:
            hits = [(k, v) for k, v in ks.items() if frag in k]
            assert hits, (src, frag, sorted(ks))
            for k, (scratch, vgpr) in hits:
                assert scratch == 0, f"{k}: {scratch} bytes of scratch per lane ({vgpr} registers) -- its arrays have left the registers"
