"""Which conv launches of the fused plan miss the fast contract, and which table is to blame (run on the GPU box)."""
import sys
import numpy as np
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from hawq_amd.api import build_quantized_resnet, calibrate
from hawq_amd.engine import IntegerEngine
from hawq_amd.quant_utils import tables_are_fast
from hawq_amd.skeleton import synthetic_images

arch, scheme = (sys.argv + ["resnet50", "uniform8"])[1:3]
model = build_quantized_resnet(arch, scheme, seed=0).cuda()
calibrate(model, synthetic_images(8, seed=0).cuda())
eng = IntegerEngine(model, use_graph=False, autotune=False, chains=1)


def report(tag, m, ek, vb):
    m, ek = np.asarray(m.cpu() if torch.is_tensor(m) else m, np.int64).reshape(-1), np.asarray(ek.cpu() if torch.is_tensor(ek) else ek, np.int64).reshape(-1)
    e, k = ek & 0xff, ek >> 8
    vb = np.broadcast_to(np.asarray(vb, np.int64), m.shape)
    tz = np.array([((int(x) & -int(x)).bit_length() - 1) if x else 0 for x in m])
    bad = ~((m == 0) | (tz <= e - 1 - k - vb)) | (e < 33) | (e > 62) | (vb + k > 31)
    print(f"    {tag}: fast={tables_are_fast(m, ek, vb)}  e {e.min()}..{e.max()}  k {k.min()}..{k.max()}  vbits {vb.min()}..{vb.max()}  "
          f"bad channels {int(bad.sum())}/{bad.size}" + (f"  first bad: m={m[bad][0]} tz={tz[bad][0]} e={e[bad][0]} k={k[bad][0]} vb={vb[bad][0]}" if bad.any() else ""))


for u in eng.P['units']:
    for i, ent in enumerate(u['convs']):
        if ent.get('fast'):
            continue
        print(f"{u['name']}.quant_convbn{i + 1}: NOT fast")
        report("main table", ent['m'], ent['e'], ent['conv'].vbits)
        if i == len(u['convs']) - 1:
            if u['resize']:
                report("identity conv table", u['m_id'], u['e_id'], u['ident'].vbits)
            else:
                report("identity scalar table", [u['m_id_s']], [u['e_id_s']], 17)
