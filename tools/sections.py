"""Diagnostic: per-section wave-cycle shares of the path kernel (needs the -DPINE_PROFILE_SECTIONS build:
   make -C pine_amd/csrc OUT=../lib/libpine_gpu_prof.so EXTRA=-DPINE_PROFILE_SECTIONS;
   PINE_GPU_LIB=pine_amd/lib/libpine_gpu_prof.so python tools/sections.py [c2|c2readme|c4|c5])"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pine_amd
from pine_amd import scenes, _lib
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
if cfg in ("c2", "committed"): scene, spp, depth = scenes.cbox((640, 640), "committed"), 256, 8
elif cfg in ("c2readme", "readme"): scene, spp, depth = scenes.cbox((640, 640), "readme"), 256, 8
elif cfg == "c4": scene, spp, depth = scenes.classic_cones((720, 360), 100), 64, 6
elif cfg == "c5": scene, spp, depth = scenes.sss((640, 640), 3), 512, 8
else: raise SystemExit("config: c2 | c2readme | c4 | c5")
w, h = scene.camera.film().size
plan = pine_amd.Plan(scene, spp, depth, timing=True)
film = torch.zeros((h, w, 4), device="cuda")
for _ in range(2):
    plan.launch(film.data_ptr(), torch.cuda.current_stream().cuda_stream)
st = plan.stats()
queue = st.block_threads == 1024
names = ["loop", "regen", "trav_closest", "surface+terminal", "sample_bxdf", "light_sample", "trav_shadow",
         "nee_eval", "bsdf_sample+push", "fold+store"]
if queue:
    names = ["pick+pop", "S:load+surface", "S:sampler+light", "S:shadow trav", "S:nee eval", "bsdf+fold store / camera",
             "closest trav", "T:result+fold", "T:items", "state store", "push", "idle", "W:walk step", "W:store"]
out = (C.c_uint64 * 16)()
_lib.check(_lib.lib.pine_gpu_plan_debug_sections(plan._h, out))
tot = max(1, sum(out[:len(names)]))
print(f"{cfg}: {'queue' if queue else 'mega'} kernel  trace_ms {st.trace_ms:.2f} vertices/sample {st.vertices/st.camera_samples:.3f} walk steps/sample {st.walk_steps/st.camera_samples:.3f} "
      f"lds {st.lds_bytes} grid {st.grid_blocks} accel_build_ms {st.accel_build_ms:.2f} upload_ms {st.upload_ms:.2f}")
for n, v in zip(names, out):
    print(f"  {n:26s} {v/tot*100:6.2f}%")
