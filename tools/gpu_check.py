import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import oracle_lib as O
from realtimepathtracingresearchframework_amd import abi, backend, scenes
np.set_printoptions(precision=6, suppress=True)

def compare(scene, W, H, spp, variant):
    r = backend.RenderHip()
    r.initialize(W, H); r.set_scene(scene)
    cfg = backend.RenderConfiguration(scene.camera_params(), active_variant=variant, reset_accumulation=True)
    st = r.render(cfg, spp=spp, count_traversal=True)
    img = np.zeros((H,W,4), np.float32); r.readback_framebuffer(img)
    osc = O.OracleScene(scene)
    nodes,tris,insts = r.export_bvh()
    osc.import_bvh(nodes,tris,insts)
    ref, ost = osc.render(W,H,spp,variant=variant, bvh_mode=O.BVH_IMPORTED, count=True)
    d = img[...,:3]-ref[...,:3]
    rmse = np.sqrt(np.mean(d**2))
    print(scene.name, "variant",variant,"rmse",rmse,"maxabs",np.abs(d).max(),"frac>1e-4",(np.abs(d).max(axis=2)>1e-4).mean(), "exact frac", (d==0).all(axis=2).mean())
    print("  gpu rays",st.raw.rays_closest,st.raw.rays_shadow,"nodes",st.raw.nodes_visited,"tris",st.raw.tris_tested,"ms",st.render_time)
    print("  cpu rays",ost.rays_closest,ost.rays_shadow,"nodes",ost.nodes_closest+ost.nodes_shadow,"tris",ost.tris_closest+ost.tris_shadow)
    print("  alpha equal", np.array_equal(img[...,3], ref[...,3]), "nan", np.isnan(img).sum())
    r.close()
    return img, ref

s = scenes.cornell32()
compare(s,128,128,2,abi.VARIANT_GLTF)
compare(s,128,128,2,abi.VARIANT_SIMPLE)
t = scenes.two_level_test()
compare(t,160,120,2,abi.VARIANT_GLTF)
g = scenes.grid(200,100)
compare(g,320,180,2,abi.VARIANT_SIMPLE)
compare(g,320,180,2,abi.VARIANT_GLTF)
# trace parity
r = backend.RenderHip(); r.initialize(64,64); r.set_scene(t)
rng = np.random.default_rng(1)
n=20000
q = np.zeros((n,8),np.float32); q[:,0:3]=rng.uniform(-6,6,(n,3)); dd=rng.normal(size=(n,3)); dd/=np.linalg.norm(dd,axis=1,keepdims=True); q[:,4:7]=dd; q[:,7]=1e20
q[:,3] = np.zeros(n,np.int32).view(np.float32)
res = r.render_ray_queries(q)
osc = O.OracleScene(t); ref = osc.trace(q, bvh_mode=O.BVH_BRUTE)
print("trace bit-equal vs brute:", np.array_equal(res.view(np.uint32), ref.view(np.uint32)), "hits", (res[:,0]>=0).sum())
