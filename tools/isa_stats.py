#!/usr/bin/env python3
"""Static instruction statistics of the gfx950 kernels, from hipcc's device assembly.

  hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -S --offload-device-only cat_amd/csrc/ntt.hip -o build/isa/ntt.s
  python3 tools/isa_stats.py build/isa/ntt.s [substring ...]

Per kernel: VGPRs, SGPRs, LDS bytes, waves/SIMD by registers, and instruction counts by class over the kernel body
(static counts: a loop body counts once; the column kernels are straight-line code, the streaming row kernels are one
loop whose body is found between the labels of its backward branch).  VALU classes follow the issue costs measured on
MI355X (tools/ubench_isa.hip): `mul` = v_mad_u64_u32 / v_mul_lo/hi_u32, `vop64` = 64-bit adds, compares, carry ops and
selects, `vop2` = the plain 32-bit ops that issue twice as fast.
"""
import re
import sys
from collections import Counter, OrderedDict


def demangle_hint(name):
    m = re.match(r"_ZN\d+_GLOBAL__N_1(\d+)", name)
    if not m:
        return name
    i = m.end() - len(m.group(1))
    n = int(m.group(1))
    j = i + len(m.group(1))
    return name[j:j + n] + name[j + n:][:60]


FAST_VOP2 = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshlrev_b32",
             "v_lshrrev_b32", "v_ashrrev_i32", "v_accvgpr", "v_add3_u32", "v_lshl_add_u32", "v_lshl_or_b32", "v_and_or_b32",
             "v_or3_b32", "v_xad_u32", "v_bfe_u32", "v_add_lshl_u32", "v_not_b32", "v_bfi_b32")


def classify(op):
    if op.startswith(("v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_i64_i32", "v_mul_u32_u24", "v_mad_u32_u24",
                      "v_mul_hi_u32_u24")):
        return "valu_mul"
    if op.startswith(("v_cndmask",)):
        return "valu_select"
    if op.startswith(("v_cmp", "v_cmpx")):
        return "valu_cmp"
    if op.startswith(("v_lshl_add_u64", "v_add_co", "v_addc_co", "v_sub_co", "v_subb_co", "v_subrev_co", "v_subbrev_co",
                      "v_lshlrev_b64", "v_lshrrev_b64", "v_add_u64")):
        return "valu_add64"
    if op.startswith(FAST_VOP2):
        return "valu_vop2"
    if op.startswith("v_readfirstlane") or op.startswith("v_readlane") or op.startswith("v_writelane"):
        return "valu_lane"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "vmem_load"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store")):
        return "vmem_store"
    if op.startswith(("global_atomic", "buffer_atomic", "flat_atomic")):
        return "vmem_atomic"
    if op.startswith("ds_read") or op.startswith("ds_load") or op.startswith("ds_bpermute") or op.startswith("ds_permute"):
        return "lds_read"
    if op.startswith("ds_write") or op.startswith("ds_store"):
        return "lds_write"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def parse(path):
    kernels = OrderedDict()
    cur = None
    meta = {}
    with open(path) as f:
        for ln in f:
            m = re.match(r"^(_Z\w+):\s*(;.*)?$", ln)
            if m:
                cur = m.group(1)
                kernels[cur] = []
                continue
            if cur is None:
                continue
            s = ln.strip()
            if s.startswith(".end_amdhsa_kernel") or s.startswith(".Lfunc_end"):
                cur = None if s.startswith(".Lfunc_end") else cur
                continue
            if s.startswith("; ") or not s or s.startswith("."):
                if s.startswith(".L") and s.rstrip().endswith(":"):
                    kernels[cur].append(("label", s.rstrip(":")))
                continue
            op = s.split()[0]
            arg = s[len(op):].split(";")[0].strip()
            kernels[cur].append((op, arg))
    # resource usage from the .amdhsa_ directives / comment block
    text = open(path).read()
    for name in kernels:
        m = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"\n(.*?)\.end_amdhsa_kernel", text, re.S)
        info = {}
        if m:
            blk = m.group(1)
            for key in ("next_free_vgpr", "next_free_sgpr", "group_segment_fixed_size", "private_segment_fixed_size",
                        "accum_offset"):
                mm = re.search(r"\.amdhsa_" + key + r"\s+(\d+)", blk)
                if mm:
                    info[key] = int(mm.group(1))
        meta[name] = info
    return kernels, meta


def waves_per_simd(vgpr):
    alloc = -(-vgpr // 8) * 8
    return min(8, 512 // max(alloc, 8)), alloc


def loop_body(ins):
    """largest span between a label and a backward branch to it (the streaming loop), or None"""
    labels = {a: i for i, (o, a) in enumerate(ins) if o == "label"}
    best = None
    for i, (o, a) in enumerate(ins):
        if o.startswith(("s_cbranch", "s_branch")) and a in labels and labels[a] < i:
            span = (labels[a], i)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    return best


def main():
    path = sys.argv[1]
    filt = sys.argv[2:]
    kernels, meta = parse(path)
    for name, ins in kernels.items():
        if not meta[name]:
            continue
        short = demangle_hint(name)
        if filt and not any(f in name for f in filt):
            continue
        body = [(o, a) for o, a in ins if o != "label"]
        c = Counter(classify(o) for o, _ in body)
        valu = sum(v for k, v in c.items() if k.startswith("valu"))
        info = meta[name]
        w, alloc = waves_per_simd(info.get("next_free_vgpr", 0))
        print(f"== {short}\n   {name}")
        print(f"   vgpr {info.get('next_free_vgpr')} (alloc {alloc}, {w} waves/SIMD)  sgpr {info.get('next_free_sgpr')}  "
              f"lds {info.get('group_segment_fixed_size')} B  scratch {info.get('private_segment_fixed_size')} B")
        print(f"   instructions {len(body)}  VALU {valu}: " + ", ".join(f"{k[5:]} {c[k]}" for k in sorted(c) if k.startswith("valu")))
        print("   other: " + ", ".join(f"{k} {c[k]}" for k in sorted(c) if not k.startswith("valu")))
        lb = loop_body(ins)
        if lb:
            lbody = [(o, a) for o, a in ins[lb[0]:lb[1] + 1] if o != "label"]
            lc = Counter(classify(o) for o, _ in lbody)
            lv = sum(v for k, v in lc.items() if k.startswith("valu"))
            print(f"   main loop: {len(lbody)} instructions, VALU {lv}: " +
                  ", ".join(f"{k[5:]} {lc[k]}" for k in sorted(lc) if k.startswith("valu")) +
                  "; " + ", ".join(f"{k} {lc[k]}" for k in sorted(lc) if not k.startswith("valu")))


if __name__ == "__main__":
    main()
