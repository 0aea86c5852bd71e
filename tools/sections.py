"""Diagnostic: per-section wave-cycle shares of the path kernel (needs the -DPINE_PROFILE_SECTIONS build:
   make -C pine_amd/csrc OUT=../lib/libpine_gpu_prof.so EXTRA=-DPINE_PROFILE_SECTIONS;
   PINE_GPU_LIB=pine_amd/lib/libpine_gpu_prof.so python tools/sections.py)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pine_amd
from pine_amd import scenes, _lib
names = ["loop", "regen", "trav_closest", "surface+terminal", "sample_bxdf", "light_sample", "trav_shadow",
         "nee_eval", "bsdf_sample+push", "fold+store"]
if os.environ.get("PINE_GPU_KERNEL") == "queue":
    names = ["pick+pop", "S:load+surface", "S:sampler+light", "S:shadow trav", "S:nee eval", "bsdf+fold store / camera",
             "closest trav", "T:result+fold", "T:items", "state store", "push", "idle"]
cam = sys.argv[1] if len(sys.argv) > 1 else "committed"
scene = scenes.cbox((640, 640), cam)
plan = pine_amd.Plan(scene, 256, 8, timing=True)
film = torch.zeros((640, 640, 4), device="cuda")
for _ in range(2):
    plan.launch(film.data_ptr(), torch.cuda.current_stream().cuda_stream)
st = plan.stats()
out = (C.c_uint64 * 16)()
_lib.check(_lib.lib.pine_gpu_plan_debug_sections(plan._h, out))
tot = sum(out)
print(f"trace_ms {st.trace_ms:.2f} vertices/sample {st.vertices/st.camera_samples:.3f}")
for n, v in zip(names, out):
    print(f"  {n:18s} {v/tot*100:6.2f}%")
