for r in 1 2; do for v in product p4r3 p4r2; do
 if [ $v = product ]; then unset BNN_HIP_LIB; else export BNN_HIP_LIB=$PWD/bayesian-neural-network_amd/bnn_hip/libbnn_hip_$v.so; fi
 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value']), d['ms_per_step'], d['roofline'].get('avg_launch_us'))"
done; done
