import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from autourdf_amd.engine import SequenceRegistrar
from autourdf_amd.synthetic import make_sequence, initial_segmentation
dev = torch.device("cuda:0")
seq = make_sequence("wx200_5", 0, 6, 4096)
mats, clusters, _ = initial_segmentation(seq[0], 20, seed=0)
for rot, hidden in (("q", 512), ("6d", 512), ("rpy", 3), ("rpy", 512)):
    r = SequenceRegistrar(mats.astype(np.float32), clusters, 4096, rot, hidden, 300, True, dev, 0)
    out = []
    for f in seq[1:5]:
        m, res = r.step(torch.as_tensor(f, dtype=torch.float64, device=dev))
        out.append([round(float(v), 5) for v in res.cpu()])
    print(rot, hidden, out, flush=True)
