import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, pine_amd
from pine_amd import scenes
from oracle import oracle
name = sys.argv[1]
sc = {'classic20': lambda: scenes.classic_cones((180, 90), 20), 'sss': lambda: scenes.sss((96,96),2)}[name]()
spp, depth = (64, 6) if name == 'classic20' else (64, 8)
film = pine_amd.PathIntegrator(pine_amd.BlueSampler(spp), depth).render(sc).pixels
ref, _ = oracle.render(sc.describe(), sc.camera.film().size, spp, depth)
print(name, 'mismatched', int((ref.view(np.uint32) != film.view(np.uint32)).any(axis=2).sum()))
