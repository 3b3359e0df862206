"""The REFERENCE's own BCNN (imported from /root/reference) and the oracle's restatement of it, timed side by side on the
host cores of the build container: one full training step (forward, CrossEntropy(label_smoothing 0.1), backward, SGD
momentum 0.9) at BASELINE.json configs[0] (batch 4, 448 x 448, 200 classes, fp32, torch CPU).  /root/reference does not
exist on the GPU box, so `bench.py`'s `cpu_baseline` there times the oracle ("kind": "port"); this script is the evidence
that the port and the reference run at the same speed where both exist.  Test infrastructure.
    python oracle/time_reference.py [threads]"""
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('HAWKEYE_REFERENCE', '/root/reference')
sys.path.insert(0, os.path.join(HERE, '_stubs'))
sys.path.insert(0, REF)
sys.path.insert(0, HERE)

threads = int(sys.argv[1]) if len(sys.argv) > 1 else min(os.cpu_count() or 1, 32)
torch.set_num_threads(threads)

import model  # noqa: E402,F401  (reference package; registers all methods)
from model.registry import MODEL  # noqa: E402
from yacs.config import CfgNode as CN  # noqa: E402  (oracle/_stubs)

import hawkeye_oracle as O  # noqa: E402

M_BCNN = sys.modules['model.methods.BCNN']
real_vgg16 = M_BCNN.vgg16
M_BCNN.vgg16 = lambda pretrained=True: real_vgg16(pretrained=False)      # no network: random init, same architecture


def step_time(m, bs=4, image=448, classes=200, steps=3):
    torch.manual_seed(0)
    opt = torch.optim.SGD(m.parameters(), lr=0.005, momentum=0.9, weight_decay=1e-5)
    crit = torch.nn.CrossEntropyLoss(label_smoothing=0.1)
    x, y = torch.randn(bs, 3, image, image), torch.randint(0, classes, (bs,))
    best = None
    for it in range(steps + 1):
        t0 = time.time()
        loss = crit(m(x), y)
        opt.zero_grad()
        loss.backward()
        opt.step()
        dt = time.time() - t0
        if it > 0:
            best = dt if best is None else min(best, dt)
    return bs / best


ref = MODEL.get('BCNN')(CN(dict(stage=2, num_classes=200)))
port = O.BCNNOracle(200, stage=2)
out = {'threads': threads, 'host_cores': os.cpu_count(), 'workload': 'BCNN stage-2 train step, batch 4, 448x448, best of 3 after 1 warm-up',
       'reference_images_per_sec': round(step_time(ref), 3), 'oracle_port_images_per_sec': round(step_time(port), 3)}
print(json.dumps(out))
