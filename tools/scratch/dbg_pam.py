import sys; sys.path.insert(0,'.')
import numpy as np, torch
from waveforms_amd.link import SOQPSKLink
for off in (-4,0,3):
    ref = SOQPSKLink(70001, 8, fuse=7, timing_offset=off, detector="PAM")
    fus = SOQPSKLink(70001, 8, fuse=15, timing_offset=off, detector="PAM")
    for l in (ref,fus): l.run_block(4.0, seed=7, stream_id=9, skip_bits=55)
    lr, lf = ref.layout(), fus.layout(); calls=lr["calls"]
    a = ref.workspace[lr["off_mf"]:lr["off_mf"] + calls * 32].view(torch.float64).reshape(calls, 4).cpu().numpy()
    b = fus.workspace[lf["off_mf"]:lf["off_mf"] + calls * 32].view(torch.float64).reshape(calls, 4).cpu().numpy()
    bad = np.nonzero(np.abs(a-b).max(axis=1) > 2e-12)[0]
    print(off, calls, len(bad), bad[:40], bad[-10:])
    print(" mod 1024:", sorted(set((bad % 1024).tolist()))[:40])
    print(" mod 64:", sorted(set((bad % 64).tolist())))
    print(" slots bad:", [(np.abs(a-b)[:,j] > 2e-12).sum() for j in range(4)])
