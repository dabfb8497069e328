import sys, numpy as np
sys.path.insert(0, '.')
import oracle
from waveforms_amd import _hip, device as dev
n=400_000
bits,_=oracle.glfsr_bits(0x420000,0x7FFFFF,n)
noise=oracle.philox_awgn(oracle.sigma_for_ebn0(0.0,8),7,42,0,(n+1)*8)
res=oracle.detection_run(bits, oracle.freq_pulse_soqpsk_tg(8),0.25,8,None,noise=noise,length=2)
rows=_hip.to_device(np.ascontiguousarray(res["mf_rows"]))
for k in (dev.viterbi_unmerged, dev.viterbi_repaired, dev.viterbi_cascaded): k(reset=True)
for w in (2,2,4,16):
    b,s=dev.viterbi_detect(rows, warmup=w)
    print('warmup',w,'unmerged',dev.viterbi_unmerged(reset=True),'repaired',dev.viterbi_repaired(reset=True),'cascaded',dev.viterbi_cascaded(reset=True), 'ok', np.array_equal(_hip.to_host(b),res["det_bits"]))
