# A/B of builds of the persistent window kernel at the reference's default size, 1000-step calls, median of 5:
#   bash tools/ab_window.sh LOGNAME VARIANT...     VARIANT = a tools/ab_build.py name, or "shipped" for the library as built
# (log: gpurun_out/r06ad/LOGNAME.log; what the variants of round 6 read: profiles/r06_window_kernel.md, sections (e)-(f))
mkdir -p gpurun_out/r06ad
out=gpurun_out/r06ad/$1.log; shift
: > $out
for v in "$@"; do
  echo "--- $v" >> $out
  if [ "$v" = shipped ]; then timeout -k 10 100 python tools/sweep.py --rows 1080 --cols 1920 --steps 1000 kernel=6 >> $out 2>&1
  else GS_HIP_LIBRARY=grayscott_amd/variants/libgs_hip_$v.so timeout -k 10 100 python tools/sweep.py --rows 1080 --cols 1920 --steps 1000 kernel=6 >> $out 2>&1; fi
done
grep -- "---\|median" $out
