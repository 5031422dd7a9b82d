R=$PWD; O=$R/gpurun_out/knn4; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 3; do
  if [ $v = 0 ]; then unset DANCE_HIP_LIB; else export DANCE_HIP_LIB=$R/dance_amd/libdancehip_knnabl$v.so; fi
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/t$v -o t -- python $R/scripts/knn_one.py 1000000 50 2 > $O/t$v.log 2>&1
  echo "== variant $v"; python $R/scripts/rocpd_stats.py $(find $O/t$v -name "*.db" | head -1) $O/knn_abl$v | grep -E "fold_filter|rerank|sreg" | cut -c1-60,110-170
done
find $O -name "*.db" -delete
