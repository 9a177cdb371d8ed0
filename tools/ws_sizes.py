import sys, ctypes; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from adaptigraph_amd import _lib, configs
from adaptigraph_amd.model import DynamicsPredictor
w = dict(np.load("/root/repo/tests/golden/weights_seed0.npz"))
m = DynamicsPredictor(configs.model_config(), configs.material_config("rope"), configs.dataset_config("rope"), "cuda:0")
m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.to("cuda:0").eval()
L, h = _lib.lib(), m.handle(torch.device("cuda:0"))
for B, N in ((256, 1001), (1024, 1001), (20000, 201), (500, 201)):
    prm = _lib.RolloutParams(B, N, N - 1, 1, 10, 0, 1, 10, 0, 0.0)
    out = []
    for ss in (0, 1):
        m.set_option("shared_state", ss)
        out.append(L.ag_rollout_workspace_bytes_for(h, ctypes.byref(prm)) / 1e9)
    print(B, N, "plain %.2f GB  shared %.2f GB" % tuple(out))
