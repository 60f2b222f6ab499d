#!/usr/bin/env python3
"""(CPU) What does the three-comparison child order (csrc/dtraverse.h) cost in node visits against a full sort by entry distance?
Builds a scene's tree on the host, lets the oracle walk it for the rays of a small frame in both orders (each in its own process:
the switch is read once) and prints nodes / triangles per ray.   tools/order_probe.py [scene ...]   (default: grid_1m; also forest = C4's 10 M triangles, small_forest; RPTR_FLATTEN=1 for the flattened tree)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import oracle_lib as O
    from realtimepathtracingresearchframework_amd import abi, backend, scenes
    name = sys.argv[2]
    s = scenes.forest(n_meshes=6, tris_per_tree=8000, n_instances=150, name="forest") if name == "small_forest" else getattr(scenes, name)()
    nodes, tris, insts, _ = backend.build_bvh_host(s)
    osc = O.OracleScene(s)
    osc.import_bvh(nodes, tris, insts)
    _, st = osc.render(480, 270, 1, variant=abi.VARIANT_SIMPLE, bvh_mode=O.BVH_IMPORTED, count=True)
    print("%-16s %-8s %8d nodes | closest-hit rays: %.2f nodes %.2f triangles per ray | shadow rays: %.2f nodes %.2f triangles per ray" % (
        name, ("sorted" if os.environ.get("ORC_SORT_BY_DISTANCE") else "pairs") + (" key " + os.environ["ORC_ORDER_KEY"] if os.environ.get("ORC_ORDER_KEY") else ""), len(nodes) // 16, st.nodes_closest / st.rays_closest,
        st.tris_closest / st.rays_closest, st.nodes_shadow / max(1, st.rays_shadow), st.tris_shadow / max(1, st.rays_shadow)))
    sys.exit(0)

for scene in (sys.argv[1:] or ["grid_1m"]):
    for sort in ((False,) if os.environ.get("ORC_ORDER_KEY") else (False, True)):
        env = dict(os.environ)
        env.pop("ORC_SORT_BY_DISTANCE", None)
        if sort:
            env["ORC_SORT_BY_DISTANCE"] = "1"
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", scene], env=env)
