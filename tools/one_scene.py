"""Diagnostic: render one named test scene on the GPU (kernel chosen by PINE_GPU_KERNEL) and compare with the oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pine_amd
from pine_amd import scenes
from oracle import oracle
name = sys.argv[1]
sc, spp, depth = {
    'classic12': lambda: (scenes.classic_cones((90, 45), 12), 32, 6),
    'classic20': lambda: (scenes.classic_cones((180, 90), 20), 64, 6),
    'zoo': lambda: (scenes.shapes_zoo((48, 48)), 16, 5),
    'sss': lambda: (scenes.sss((96, 96), 2), 64, 8),
}[name]()
w, h = sc.camera.film().size
print(name, 'kernel env', os.environ.get('PINE_GPU_KERNEL'), flush=True)
plan = pine_amd.Plan(sc, spp, depth, timing=True)
film = torch.full((h, w, 4), -1.0, device='cuda')
t0 = time.time()
plan.launch(film.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
st = plan.stats()
print(f'  launched+synced in {time.time()-t0:.2f}s threads/block={st.block_threads} grid={st.grid_blocks} trace_ms={st.trace_ms:.2f}', flush=True)
ref, _ = oracle.render(sc.describe(), (w, h), spp, depth)
f = film.cpu().numpy()
print('  mismatched', int((ref.view(np.uint32) != f.view(np.uint32)).any(axis=2).sum()), flush=True)
