#!/usr/bin/env python3
"""(CPU) What would a sort-free child order cost in node visits?  Builds the 1 M-triangle tree of C2 on the host, lets the oracle walk it
for the rays of a small frame, and compares: the SAH-driven collapse + children ordered by entry distance (what is built) against a
balanced collapse (LL, LR, RL, RR) ordered by distance, and the balanced collapse ordered by ray-direction signs (no sort network).
Run each configuration in its own process: tools/order_probe.py <mode>  with mode in base | balanced | signs."""
import os
import sys

mode = sys.argv[1] if len(sys.argv) > 1 else "base"
if mode in ("balanced", "signs"):
    os.environ["RPTR_COLLAPSE_BALANCED"] = "1"
if mode == "signs":
    os.environ["ORC_ORDER_EXPERIMENT"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle_lib as O  # noqa: E402
from realtimepathtracingresearchframework_amd import abi, backend, scenes  # noqa: E402

s = scenes.grid_1m() if len(sys.argv) < 3 else getattr(scenes, sys.argv[2])()
nodes, tris, insts, need = backend.build_bvh_host(s)
osc = O.OracleScene(s)
osc.import_bvh(nodes, tris, insts)
W, H = 480, 270
_, st = osc.render(W, H, 1, variant=abi.VARIANT_SIMPLE, bvh_mode=O.BVH_IMPORTED, count=True)
print("%-9s nodes %8d  | closest: %.2f nodes %.2f tris per ray (%d rays) | shadow: %.2f nodes %.2f tris per ray (%d rays)" % (
    mode, len(nodes) // 16, st.nodes_closest / st.rays_closest, st.tris_closest / st.rays_closest, st.rays_closest,
    st.nodes_shadow / max(1, st.rays_shadow), st.tris_shadow / max(1, st.rays_shadow), st.rays_shadow))
