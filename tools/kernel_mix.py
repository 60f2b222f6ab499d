"""Static VALU instruction mix of the path kernels (disassembly of the built library), by issue class, for the issue-rate microbenchmark:
tools/kernel_mix.py [lib.so] [--flags] -> profiles/r05_kernel_mix.json; --flags prints the -D options tools/microbench/valu_issue.hip is
compiled with so that it measures the issue ceiling of EACH kernel's own mix (the shade kernel was held against the
traversal's ceiling and came out above 1)."""
import json, os, re, subprocess, sys, collections
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
LIB = next((a for a in sys.argv[1:] if a.endswith(".so")), os.path.join(ROOT, "realtimepathtracingresearchframework_amd", "librptr_hip.so"))
# (label, substring of the mangled kernel name): the benchmarked instantiations of C2 (Lambert, one instance) and C3 (glTF + lights)
KERNELS = [
    ("extend_first", "_Z11rp_k_extendILb0ELb1ELb0ELb1ELb0EE"), ("extend", "_Z11rp_k_extendILb0ELb0ELb0ELb1ELb0EE"),
    ("connect", "_Z12rp_k_connectILb0ELb0ELb1EE"),
    ("shade_first_lambert", "_Z10rp_k_shadeILi1ELb1ELb0ELb0ELb0ELi0EE"), ("shade_lambert", "_Z10rp_k_shadeILi1ELb0ELb0ELb0ELb0ELi0EE"),
    ("shade_first_gltf_lights", "_Z10rp_k_shadeILi0ELb1ELb1ELb0ELb0ELi0EE"), ("shade_gltf_lights", "_Z10rp_k_shadeILi0ELb0ELb1ELb0ELb0ELi0EE"),
    ("tail_lambert", "_Z9rp_k_tailILi1ELb0ELb0ELb0ELb1ELb0ELi0EE"),
    ("resolve", "_Z12rp_k_resolve"),
]
CLASSES = ["fma", "int", "pk", "cvt", "minmax", "cnd", "cmp", "trans", "mullo", "lane"]


def classify(op):
    if not op.startswith("v_"):
        return None
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")) or "_dpp" in op or op.startswith(("v_permlane", "v_mbcnt", "v_bcnt")):
        return "lane" if not op.startswith(("v_mbcnt", "v_bcnt")) else "int"
    if op.startswith("v_pk_"):
        return "pk"
    if op.startswith("v_cndmask"):
        return "cnd"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return "cmp"
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
        return "trans"
    if op.startswith(("v_mul_lo", "v_mul_hi", "v_mad_u64", "v_mad_i64", "v_mad_u32", "v_mad_i32", "v_mul_u32", "v_mul_i32")) or "f64" in op:
        return "mullo"
    if op.startswith(("v_min", "v_max", "v_med3")):
        return "minmax"
    if op.startswith(("v_cvt", "v_bfe", "v_bfi", "v_perm", "v_alignbit", "v_alignbyte", "v_ldexp", "v_frexp", "v_fract", "v_floor", "v_ceil", "v_trunc", "v_rndne",
                      "v_fma_mix", "v_sad", "v_lerp", "v_ffb", "v_cubeid", "v_div_", "v_mad_mix", "v_lshl_or", "v_lshl_add", "v_add_lshl", "v_and_or", "v_or3", "v_xad",
                      "v_add3", "v_mad_u16", "v_mad_i16")):
        return "cvt"
    if op.startswith(("v_fma", "v_mul_f", "v_add_f", "v_sub_f", "v_subrev_f", "v_mac", "v_fmac", "v_mad_f", "v_mul_legacy", "v_mad_legacy", "v_fmaak", "v_fmamk")):
        return "fma"
    return "int"   # v_add_u32 / v_sub / v_and / v_or / v_xor / shifts / v_mov / v_accvgpr ...


def disasm(lib):
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "disasm.sh"), lib, "rp_k_"], stdout=subprocess.PIPE, universal_newlines=True).stdout
    cur, per = None, collections.OrderedDict()
    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
        if m:
            cur = m.group(1)
            per[cur] = []
            continue
        t = line.split()
        if cur is not None and t:
            per[cur].append(t[0])
    return per


def main():
    per = disasm(LIB)
    doc = {"library": os.path.relpath(LIB, ROOT), "classes": CLASSES, "kernels": {}}
    rows = []
    for label, pat in KERNELS:
        name = next((k for k in per if pat in k), None)
        if name is None:
            continue
        hist, ops = collections.Counter(), collections.Counter()
        for op in per[name]:
            c = classify(op)
            if c:
                hist[c] += 1
                ops[op] += 1
        total = sum(hist.values())
        # a loop body of ~96 instructions in the kernel's proportions (largest remainder)
        body = 96
        quota = {c: hist[c] * body / max(total, 1) for c in CLASSES}
        cnt = {c: int(quota[c]) for c in CLASSES}
        for c in sorted(CLASSES, key=lambda c: quota[c] - cnt[c], reverse=True)[: body - sum(cnt.values())]:
            cnt[c] += 1
        doc["kernels"][label] = {"symbol": name, "valu_instructions": total, "all_instructions": len(per[name]), "histogram": {c: hist[c] for c in CLASSES},
                                 "mix_of_96": cnt, "top_opcodes": dict(ops.most_common(12))}
        rows.append((label, [cnt[c] for c in CLASSES]))
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(doc, open(os.path.join(ROOT, "profiles", "r05_kernel_mix.json"), "w"), indent=1)
    # tools/microbench/kmix_gen.h: one loop body per kernel, the classes interleaved evenly (a run of one class would measure that
    # class's back-to-back behaviour, e.g. v_cndmask on one mask register, not the mix)
    macro = {"fma": "I_FMA", "int": "I_ADDU", "pk": "I_PKFMA", "cvt": "I_CVT", "minmax": "I_MIN3", "cnd": "I_CND", "cmp": "I_CMP", "trans": "I_RCP",
             "mullo": "I_MULLO", "lane": "I_LANE"}
    with open(os.path.join(ROOT, "tools", "microbench", "kmix_gen.h"), "w") as f:
        f.write("// generated by tools/kernel_mix.py from %s: the static VALU mix of each path kernel as a 96-instruction loop body\n" % doc["library"])
        f.write("#define KMIX_N %d\n" % len(rows))
        f.write("static const char *kmix_names[KMIX_N] = {%s};\n" % ", ".join('"kmix:%s"' % r[0] for r in rows))
        for k, (label, cnt) in enumerate(rows):
            left, seq, acc = dict(zip(CLASSES, cnt)), [], {c: 0.0 for c in CLASSES}
            for _ in range(sum(cnt)):
                for c in CLASSES:
                    acc[c] += dict(zip(CLASSES, cnt))[c]
                c = max((c for c in CLASSES if left[c] > 0), key=lambda c: acc[c])
                acc[c] -= sum(cnt)
                left[c] -= 1
                seq.append(c)
            body = " ".join("%s(%s%d)" % (macro[c], "p" if c == "pk" else "a", i % 8) for i, c in enumerate(seq))
            f.write("#define KMIX_BODY_%d %s\n" % (k, body))
    if "--flags" in sys.argv:
        pass
    else:
        for label, k in doc["kernels"].items():
            print("%-26s VALU %5d of %5d  %s" % (label, k["valu_instructions"], k["all_instructions"], " ".join("%s %d" % (c, k["mix_of_96"][c]) for c in CLASSES)))


if __name__ == "__main__":
    main()
