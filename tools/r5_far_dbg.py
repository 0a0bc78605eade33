import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
os.environ["SFGPU_TIMING"] = "1"; os.environ["SFGPU_EM_NO_RENUMBER"] = "1"
import numpy as np, torch
import sailfish_amd as sf
import test_gpu_persist as T
m = T._far_table_with_homes()
p = T._gpu_em(sf, torch.device("cuda:0"), m["eff"], m["rowptr"], m["ids"], m["counts"], m["R"])
print(p.optimize(use_vbem=False, tol=0.0, min_iter=0, max_iter=3))
