# the persistent window kernel on grids below its round-5 threshold: steps per exchange (fuse_steps) and window height
mkdir -p gpurun_out/r06ah
out=gpurun_out/r06ah/window_small_grids_k.log; : > $out
for sz in "512 1024" "720 1280" "1024 1024" "800 1600" "900 1600" "1000 1500"; do
  set -- $sz
  for st in 64 1000; do
    echo "--- $1 x $2, $st steps" >> $out
    timeout -k 10 100 python tools/sweep.py --rows $1 --cols $2 --steps $st kernel=0 kernel=6 kernel=6,fuse_steps=6 kernel=6,fuse_steps=8 2>&1 | grep median >> $out
    GS_HIP_WINDOW_WAVES=12,12,12 timeout -k 10 100 python tools/sweep.py --rows $1 --cols $2 --steps $st kernel=6,fuse_steps=6 kernel=6,fuse_steps=8 2>&1 | grep median | sed 's/^/waves 12: /' >> $out
  done
done
cat $out
