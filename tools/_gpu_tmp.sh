set -e
timeout -k 10 300 python tools/fuzz_front_end.py --seconds 60 --seed 5 > gpurun_out/r04_o_fuzz.log 2>&1 || { tail -20 gpurun_out/r04_o_fuzz.log; exit 1; }
tail -2 gpurun_out/r04_o_fuzz.log
