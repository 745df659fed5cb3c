mkdir -p gpurun_out/r03g
L=gpurun_out/r03g
SAENA_SETUP_TIMING=1 SAENA_SELL_NT=0 SAENA_SELLP_NT=0 timeout -k 10 400 python -m tests.perf_vcycle 256 > $L/vcycle256_nont.log 2> $L/vcycle256_nont.err
tail -16 $L/vcycle256_nont.log
SAENA_SETUP_TIMING=1 timeout -k 10 400 python -m tests.perf_vcycle 256 > $L/vcycle256_nt.log 2> $L/vcycle256_nt.err
tail -16 $L/vcycle256_nt.log
