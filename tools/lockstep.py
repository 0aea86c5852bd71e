import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pine_amd
from pine_amd import scenes
def run(name, scene, spp, depth):
    w, h = scene.camera.film().size
    plan = pine_amd.Plan(scene, spp, depth, timing=True)
    film = torch.zeros((h, w, 4), device="cuda")
    for _ in range(3):
        plan.launch(film.data_ptr(), torch.cuda.current_stream().cuda_stream)
    st = plan.stats()
    print(f"{name}: trace {st.trace_ms:.2f} ms  V={st.vertices/st.camera_samples:.3f}  ps/vertex={st.trace_ms*1e9/st.vertices:.1f}  shadow/vertex={st.shadow_rays/st.vertices:.3f}")
run("C2 committed d8", scenes.cbox((640, 640), "committed"), 256, 8)
run("readme d8", scenes.cbox((640, 640), "readme"), 256, 8)
run("readme d2 (lockstep)", scenes.cbox((640, 640), "readme"), 256, 2)
run("readme d3", scenes.cbox((640, 640), "readme"), 256, 3)
run("readme d1 (camera rays only)", scenes.cbox((640, 640), "readme"), 256, 1)
