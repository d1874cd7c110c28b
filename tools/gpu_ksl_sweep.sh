T=$PWD/bayesian-neural-network_amd/bnn_hip/libbnn_hip_tune.so
for k in 0 3 4 5 6 7 8; do
  echo "== forced slices $k (0: the plan's own)"
  if [ $k = 0 ]; then BNN_HIP_LIB=$T SWEEP_S=4,8,10,12,16,24 python tools/few_sample_sweep.py 2>&1 | grep "S=" | sed 's/| tile.*| kslice+sig\[/| kslice+sig[/; s/| auto.*//'; else BNN_HIP_LIB=$T BNN_TUNE_KSL=$k SWEEP_S=4,8,10,12,16,24 python tools/few_sample_sweep.py 2>&1 | grep "S=" | sed 's/| tile.*| kslice+sig\[/| kslice+sig[/; s/| auto.*//'; fi
done
