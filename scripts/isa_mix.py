#!/usr/bin/env python3
"""Static instruction mix of a kernel's main loop, priced with the measured issue costs of
profiles/r03_issue_rate_ubench.txt (scripts/ubench/issue_rate.hip).

  python scripts/isa_mix.py [--so grample_amd/libgrample_hip.so] [--kernel 'k_sweep_par4ILi4ELb1ELi1ELi2ELi0E'] [--json]

Why: the sweep kernels are bound by instruction issue, not by bytes.  gfx950's SIMD issues the
"simple" 32-bit VALU operations (add, sub, and, or, xor, bitop3, mov, lshrrev, f32 add/mul/fma) at one wave-instruction
per ~2.3 cycles once it holds two or more wavefronts, and everything else (multiplies, multiply-adds, compares, SDWA/DPP,
3-operand integer ops, f64, packed f32, conversions) at one per ~4.2; lane-swap permutes and transcendentals take ~8.1.
LDS instructions are priced per SIMD with all four SIMDs of the CU issuing (ds_read_u8/b32/b64 8.2, ds_write_b8 16.1,
ds_read_b128 16.2, ds_add_u64 24.2).  The roofline of such a kernel is therefore "cycles of the VALU port (or of the LDS
pipe) its instruction stream needs" against "cycles available", and needs the MIX: this script disassembles the built
library, finds the kernel's hottest loop (the backward branch with the longest span) and prices every instruction in it.
bench.py multiplies the resulting mean cost per VALU / LDS instruction with the DYNAMIC instruction counts of the rocprofv3
SQ pass (profiles/roofline_inputs.json)."""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

# cycles per wave-instruction on one SIMD, 4 wavefronts per SIMD column of profiles/r03_issue_rate_ubench.txt (the
# occupancy the headline kernel runs at), rounded; see that file for 1 / 2 / 8 wavefronts
FULL, HALF, SLOW = 2.4, 4.25, 8.12
VALU_FULL = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_bitop3_b32", "v_mov_b32", "v_lshrrev_b32",
             "v_fma_f32", "v_add_f32", "v_mul_f32", "v_sub_f32", "v_fmac_f32", "v_mac_f32", "v_not_b32", "v_add_co_u32", "v_sub_co_u32",
             "v_accvgpr_read_b32", "v_accvgpr_write_b32"}  # (the last five: not measured, priced like their class)
VALU_SLOW = {"v_permlane32_swap_b32", "v_permlane16_swap_b32", "v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64",
             "v_rcp_iflag_f32", "v_sin_f32", "v_cos_f32"}
LDS_COST = {"ds_read_u8": 8.3, "ds_read_i8": 8.3, "ds_read_u16": 8.3, "ds_read_b32": 8.3, "ds_read_b64": 8.4, "ds_read_u8_d16": 8.3, "ds_read_u8_d16_hi": 8.3,
            "ds_read_u16_d16": 8.3, "ds_read_u16_d16_hi": 8.3, "ds_read2_b32": 8.4, "ds_read_b96": 16.3, "ds_read_b128": 16.3, "ds_read2_b64": 16.3,
            "ds_write_b8": 16.3, "ds_write_b16": 16.3, "ds_write_b32": 16.3, "ds_write_b64": 16.3, "ds_write_b128": 32.0, "ds_write2_b32": 16.3,
            "ds_add_u64": 24.3, "ds_add_u32": 16.3, "ds_bpermute_b32": 8.3, "ds_swizzle_b32": 8.3}


def strip_suffix(op):
    for suf in ("_e32", "_e64", "_sdwa", "_dpp", "_e64_dpp"):
        if op.endswith(suf):
            return op[: -len(suf)], suf
    return op, ""


def price(op_full):
    """-> (class, cycles): class in valu / salu / lds / vmem / smem / branch / other"""
    op, suf = strip_suffix(op_full)
    if op.startswith("v_"):
        if op in VALU_SLOW:
            return "valu", SLOW
        if suf in ("_sdwa", "_dpp", "_e64_dpp"):
            return "valu", HALF
        if op in VALU_FULL:
            return "valu", FULL
        return "valu", HALF
    if op.startswith("ds_"):
        return "lds", LDS_COST.get(op, 16.3)
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem", 0.0
    if op.startswith(("s_load", "s_buffer_load", "s_store", "s_dcache", "s_memtime", "s_memrealtime")):
        return "smem", 0.0
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_call")):
        return "branch", 4.3
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_endpgm", "s_setprio", "s_sethalt")):
        return "other", 0.0
    if op.startswith("s_"):
        return "salu", 4.3
    return "other", 0.0


def extract_code_objects(so, workdir):
    tmp = os.path.join(workdir, os.path.basename(so))
    shutil.copy(so, tmp)
    subprocess.run([OBJDUMP, "--offloading", tmp], check=True, capture_output=True)
    return sorted(os.path.join(workdir, f) for f in os.listdir(workdir) if "amdgcn" in f)


def kernel_listing(co, pattern):
    out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    lines, name, on = [], None, False
    for ln in out.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", ln)
        if m:
            on = re.search(pattern, m.group(2)) is not None and not on and name is None
            if on:
                name = m.group(2)
            continue
        if on:
            m = re.match(r"^\s+(\S+)(.*?)//\s*([0-9A-F]+):", ln)
            if m:
                lines.append((int(m.group(3), 16), m.group(1), m.group(2).strip()))
    return name, lines


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--so", default=os.path.join(ROOT, "grample_amd", "libgrample_hip.so"))
    ap.add_argument("--kernel", default=r"k_sweep_par4ILi4ELb1ELi1ELi2ELi0E", help="regex on the mangled kernel name")
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--whole", action="store_true", help="price the whole kernel instead of its hottest loop")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as wd:
        name, lines = None, []
        for co in extract_code_objects(a.so, wd):
            name, lines = kernel_listing(co, a.kernel)
            if lines:
                break
    if not lines:
        sys.exit(f"kernel matching {a.kernel!r} not found in {a.so}")
    addr = [x[0] for x in lines]
    lo, hi = 0, len(lines) - 1
    if not a.whole:
        best = None
        for i, (ad, op, args) in enumerate(lines):
            if op.startswith(("s_cbranch", "s_branch")):
                m = re.match(r"^(\d+)", args)
                if not m:
                    continue
                off = int(m.group(1))
                if off >= 32768:  # backward: 16-bit signed dword offset from the next instruction
                    tgt = ad + 4 + (off - 65536) * 4
                    if tgt in addr and (best is None or ad - tgt > best[0]):
                        best = (ad - tgt, addr.index(tgt), i)
        if best:
            lo, hi = best[1], best[2]
    def census(short):
        """short: follow every forward s_cbranch_scc* / s_cbranch_vcc* (the sweep kernels' uniform "nothing to redo" tests: level
        passes, window limits) and fall through the s_cbranch_exec* ones (exec-masked blocks, normally executed): the path of a
        quad without dependency levels.  Otherwise every instruction of the region once."""
        tot, per_op, i = {}, {}, lo
        while i <= hi:
            ad, op, args = lines[i]
            cls, cyc = price(op)
            t = tot.setdefault(cls, [0, 0.0])
            t[0] += 1
            t[1] += cyc
            pp = per_op.setdefault((cls, op), [0, cyc])
            pp[0] += 1
            nxt = i + 1
            if short and op.startswith(("s_cbranch_scc", "s_cbranch_vcc")):
                m = re.match(r"^(\d+)", args)
                off = int(m.group(1)) if m else 65535
                tgt = ad + 4 + off * 4
                if off < 256 and tgt in addr and addr.index(tgt) <= hi:  # (longer forward jumps are loop exits, not skipped blocks)
                    nxt = addr.index(tgt)
            i = nxt
        return tot, per_op

    def table(tot):
        return {c: {"instructions": n, "port_cycles": round(cy, 1), "mean_cycles": round(cy / n, 3) if n else 0.0} for c, (n, cy) in sorted(tot.items())}
    tot, per_op = census(False)
    tot_s, _ = census(True)
    out = {"kernel": name, "region": "whole kernel" if a.whole else f"hottest loop: instructions {lo}..{hi} of {len(lines)}",
           "prices": {"valu_full_rate": FULL, "valu_half_rate": HALF, "valu_swap_or_transcendental": SLOW, "salu": 4.3,
                      "source": "profiles/r03_issue_rate_ubench.txt, 4 wavefronts per SIMD"},
           "classes": table(tot), "classes_short_path": table(tot_s),
           "top": [{"op": op, "class": c, "n": n, "cycles_each": cy} for (c, op), (n, cy) in sorted(per_op.items(), key=lambda kv: -kv[1][0] * max(kv[1][1], 0.01))[:24]]}
    if a.json:
        print(json.dumps(out))
    else:
        print(out["kernel"]); print(out["region"])
        for key in ("classes", "classes_short_path"):
            print(" ", key)
            for c, d in out[key].items():
                print(f"  {c:7s} {d['instructions']:5d} instructions  {d['port_cycles']:8.1f} port cycles  mean {d['mean_cycles']:.2f}")
        for t in out["top"]:
            print(f"    {t['n']:4d} x {t['op']:28s} {t['class']:6s} {t['cycles_each']}")
    return out


if __name__ == "__main__":
    main()
