"""Generates realtimepathtracingresearchframework_amd/csrc/dstream.h: the node / leaf phases of dtraverse.h rp_wave_trace (the tuned loop body,
copied verbatim so that arithmetic, visit order and steps per ray are those of the stage kernels) inside the item life cycle of the
streaming frame. Re-run after every change of dtraverse.h's loop body: python tools/gen_dstream.py"""
import os
import re

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CSRC = os.path.join(ROOT, "realtimepathtracingresearchframework_amd", "csrc")
src = open(os.path.join(CSRC, "dtraverse.h")).read()
a = src.index('        // node phase: keeps stepping while at least RP_NODE_MIN lanes are at an inner node')
b = src.index('        if (active && cur == RP_EXIT) {\n            done(my_i, best);')
core = src[a:b]
core = re.sub(r'#ifdef RP_PROF\n.*?#endif\n', '', core, flags=re.S)
core = core.replace('''            if (LDSTOP > 0 && uint32_t(cur) < (uint32_t)LDSTOP) {
                const float4 *lp = lds_top + (uint32_t(cur) << 2);
                n0 = lp[0];
                const float4 a1 = lp[1], a2 = lp[2], a3 = lp[3];
                n1 = make_uint4(__float_as_uint(a1.x), __float_as_uint(a1.y), __float_as_uint(a1.z), __float_as_uint(a1.w));
                n2 = make_uint4(__float_as_uint(a2.x), __float_as_uint(a2.y), __float_as_uint(a2.z), __float_as_uint(a2.w));
                n3 = make_uint2(__float_as_uint(a3.x), __float_as_uint(a3.y));
            } else {
                n0 = *reinterpret_cast<const float4 *>(np);
                n1 = *reinterpret_cast<const uint4 *>(np + 16);
                n2 = *reinterpret_cast<const uint4 *>(np + 32);
                n3 = *reinterpret_cast<const uint2 *>(np + 48);
            }''', '''            n0 = *reinterpret_cast<const float4 *>(np);
            n1 = *reinterpret_cast<const uint4 *>(np + 16);
            n2 = *reinterpret_cast<const uint4 *>(np + 32);
            n3 = *reinterpret_cast<const uint2 *>(np + 48);''')
assert not re.search(r'\bLDSTOP\b', core)
core = core.replace('if (COUNT) n_nodes++;', '').replace('if (COUNT) n_nodes += 2; // 128-byte instance record', '').replace('if (COUNT) n_tris++;', '')
assert not re.search(r'\bCOUNT\b', core), [l for l in core.splitlines() if re.search(r'\bCOUNT\b', l)]
core = core.replace('ent[k] = ANY ? gap : (hit ? tn_raw : INFINITY);', 'ent[k] = anyq ? gap : (hit ? tn_raw : INFINITY);')
core = core.replace('(ANY && any_hit)', '(anyq && any_hit)')
assert not re.search(r'\bANY\b', core), [l for l in core.splitlines() if re.search(r'\bANY\b', l)]
core = re.sub(r'#if RP_SLAB_PACKED\n((?:(?!#else|#endif).)*?)#endif\n', '', core, flags=re.S)  # (the packed form's operand pairs)
core = re.sub(r'#if RP_SLAB_PACKED\n.*?#else\n(.*?)#endif\n', r'\1', core, flags=re.S)  # the default (scalar fmas) branch of the plane distances
core = core.replace('#if RP_NODE_MIN > 1\n', '').replace('#endif\n            if (cur >= 0) {', '            if (cur >= 0) {')
assert '#endif' not in core and '#if' not in core, [l for l in core.splitlines() if l.startswith('#')]
hdr = open(os.path.join(ROOT, "tools", "dstream_head.inc")).read()
ftr = open(os.path.join(ROOT, "tools", "dstream_tail.inc")).read()
open(os.path.join(CSRC, "dstream.h"), "w").write(hdr + core + ftr)
print("dstream.h: %d lines of dtraverse.h's loop body" % len(core.splitlines()))
