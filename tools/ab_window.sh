mkdir -p gpurun_out/r06ad
out=gpurun_out/r06ad/$1.log; shift
: > $out
for v in "$@"; do
  echo "--- $v" >> $out
  if [ "$v" = shipped ]; then timeout -k 10 100 python tools/sweep.py --rows 1080 --cols 1920 --steps 1000 kernel=6 >> $out 2>&1
  else GS_HIP_LIBRARY=grayscott_amd/variants/libgs_hip_$v.so timeout -k 10 100 python tools/sweep.py --rows 1080 --cols 1920 --steps 1000 kernel=6 >> $out 2>&1; fi
done
grep -- "---\|median" $out
