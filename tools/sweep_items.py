import os, sys, json, subprocess
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
def run(name, waves, ept):
    env = dict(os.environ, VS_BA_ITEM_WAVES=str(waves), VS_BA_ITEM_EPT=str(ept))
    code = f"""
import sys; sys.path.insert(0,'.')
import numpy as np, torch
from bench import make_problem
sc, ba = make_problem('{name}', 0, 1)
dc = np.full(6*sc.Nc, 1e-4); dp = np.full(3*sc.Np, 1e-4)
ms = ba.profile_linearize(dc, dp, iters=50)
print('{name} waves={waves} ept={ept}', [round(float(x)*1e3,1) for x in ms])
"""
    subprocess.run([sys.executable, '-c', code], env=env)
for name in sys.argv[1:]:
    for waves in (1, 4):
        for ept in (2, 3, 4, 5, 6, 8):
            run(name, waves, ept)
