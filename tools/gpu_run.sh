#!/bin/bash
# One parameterised script for the GPU box (run through gpurun), instead of a one-off script per run:
#   tools/gpu_run.sh TAG STAGE [STAGE ...]
# Logs and JSON lines land under gpurun_out/TAG/ (scratch; what is to be judged is copied to profiles/).
# A failing stage ends the call (no GPU step is started after a failed or timed-out one).
# Stages:
#   probe           what the GPU box has: Rust toolchain?, CPU share, rocm-smi's energy counter and power cap
#   smoke           __graft_entry__.build() + smoke(), as the driver runs them
#   newtests        this round's new GPU tests first (fast failure)
#   tests           the whole GPU suite
#   bench           bench.py as the driver runs it (--steps 20 --warmup 5) and with its defaults
#   timeline:RxC[:key=value,...]   tools/wave_timeline.py on the diagnostic build (variants/libgs_hip_trace.so)
#   sweep:RxC:STEPS:VARIANT;VARIANT...   tools/sweep.py (VARIANT = key=value,key=value)
#   libsweep:NAME:RxC:STEPS:VARIANT;...  the same against grayscott_amd/variants/libgs_hip_NAME.so
#   profile:TAG[:bench extra args]       tools/profile_gpu.sh
#   kprofile:TAG:TOOL,ARG,ARG...         tools/profile_kernel.sh (rocprofv3 passes over a python tool, e.g. tools/run_steps.py)
#   energyprof      rocprofv3 PMC passes of tools/energy_table.py --profile (tools/summarize_energy.py reads them)
#   configs         tools/baseline_configs.py
#   criterion       tools/criterion_grid.py
#   rehearsal       tools/rehearsal.sh
#   py:SCRIPT[:args]  any python tool of this repository
#   pytest:FILE[,FILE...][:K+EXPRESSION]   some GPU test files (fast failure before a long run; + stands for a blank in -k)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
for stage in "$@"; do
  name=${stage%%:*}
  rest=${stage#*:}; [ "$rest" = "$stage" ] && rest=""
  echo "=== stage $stage ($(date +%T))"
  case $name in
    probe)
      # SURVEY 8(d) / BASELINE.md row C: is there a Rust toolchain (with an offline registry) on the GPU box?
      { date -u; echo "--- command -v cargo rustc rustup"; command -v cargo rustc rustup || echo "none on PATH";
        echo "--- ls ~/.cargo ~/.rustup /usr/local/cargo /opt/rust*"; ls -d ~/.cargo ~/.cargo/registry ~/.rustup /usr/local/cargo /usr/local/rustup /opt/rust* 2>&1;
        echo "--- find / -name cargo -o -name rustc (maxdepth 4)"; find / -maxdepth 4 \( -name cargo -o -name rustc \) -not -path '/proc/*' 2>/dev/null | head; echo "(end)";
        echo "--- nproc / cpu quota / memory"; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; grep MemTotal /proc/meminfo;
        echo "--- rocm-smi energy / power cap"; rocm-smi --showenergycounter --showmaxpower --showpower --showclocks 2>&1 | head -40; } > "$OUT/probe.txt" 2>&1; rc=0; cat "$OUT/probe.txt" ;;
    smoke)
      timeout -k 10 600 python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1; rc=$?; tail -4 "$OUT/smoke.log" ;;
    newtests)
      timeout -k 10 1100 python -m pytest tests/test_gpu_bench_rehearsal.py tests/test_gpu_timed_sizes.py tests/test_gpu_baseline_configs.py -m gpu -x -q \
        -k "rehearsal or stalled or config3 or developing" \
        > "$OUT/newtests.log" 2>&1; rc=$?; tail -15 "$OUT/newtests.log" ;;
    tests)
      timeout -k 10 1100 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; rc=$?; tail -8 "$OUT/pytest_gpu.log" ;;
    bench)
      timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_style.json" 2> "$OUT/bench_driver_style.err"; rc=$?
      tail -c 3000 "$OUT/bench_driver_style.json"; tail -3 "$OUT/bench_driver_style.err"
      if [ $rc -eq 0 ]; then
        timeout -k 10 600 python bench.py > "$OUT/bench_unprofiled.json" 2> "$OUT/bench_unprofiled.err"; rc=$?
        tail -c 3000 "$OUT/bench_unprofiled.json"
      fi ;;
    timeline)
      IFS=: read -r grid kv <<< "$rest"
      GS_HIP_LIBRARY=$ROOT/grayscott_amd/variants/libgs_hip_trace.so timeout -k 10 300 python tools/wave_timeline.py ${grid%x*} ${grid#*x} ${kv//,/ } \
        > "$OUT/timeline_${grid}_${kv//[=,]/_}.log" 2>&1; rc=$?; cat "$OUT/timeline_${grid}_${kv//[=,]/_}.log" ;;
    sweep)
      IFS=: read -r grid steps variants <<< "$rest"
      timeout -k 10 600 python tools/sweep.py --rows ${grid%x*} --cols ${grid#*x} --steps "$steps" --rounds 5 ${variants//;/ } 2>&1 | tee -a "$OUT/sweep.log"; rc=$? ;;
    libsweep)
      IFS=: read -r lib grid steps variants <<< "$rest"
      echo "--- library variant $lib" | tee -a "$OUT/sweep.log"
      GS_HIP_LIBRARY=$ROOT/grayscott_amd/variants/libgs_hip_$lib.so timeout -k 10 600 python tools/sweep.py --rows ${grid%x*} --cols ${grid#*x} --steps "$steps" --rounds 5 ${variants//;/ } 2>&1 | tee -a "$OUT/sweep.log"; rc=$? ;;
    profile)
      IFS=: read -r ptag extra <<< "$rest"
      GS_BENCH_EXTRA="$extra" timeout -k 10 1100 bash tools/profile_gpu.sh "$ptag" 400 > "$OUT/profile_$ptag.log" 2>&1; rc=$?; tail -5 "$OUT/profile_$ptag.log" ;;
    sqdetail)
      # sqdetail:TAG[:bench args]   extra SQ counter passes over bench.py's headline launches (tools/profile_sq_detail.sh)
      IFS=: read -r ptag extra <<< "$rest"
      timeout -k 10 1100 bash tools/profile_sq_detail.sh "$ptag" $extra > "$OUT/sqdetail_$ptag.log" 2>&1; rc=$?; tail -30 "$OUT/sqdetail_$ptag.log" ;;
    kprofile)
      # kprofile:TAG:tools/run_steps.py,--rows,1080,...   rocprofv3 passes over a python tool (tools/profile_kernel.sh)
      IFS=: read -r ptag prog <<< "$rest"
      timeout -k 10 1100 bash tools/profile_kernel.sh "$ptag" ${prog//,/ } > "$OUT/kprofile_$ptag.log" 2>&1; rc=$?; tail -6 "$OUT/kprofile_$ptag.log" ;;
    energyprof)
      # per-flavour PMC counters of tools/energy_table.py (--profile: 40 steps per flavour), one pass per input
      rc=0
      for data in new developed; do
        ( cd /tmp && TMPDIR=/tmp timeout -k 10 1100 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU \
            --output-format csv -d "$OUT/energy_$data" -o energy -- python3 "$ROOT/tools/energy_table.py" --profile --data $data \
            > "$OUT/energyprof_$data.log" 2>&1 ) || rc=$?
        tail -3 "$OUT/energyprof_$data.log"
        [ $rc -eq 0 ] || break
      done ;;
    configs)
      timeout -k 10 1100 python tools/baseline_configs.py > "$OUT/baseline_configs.log" 2>&1; rc=$?; tail -30 "$OUT/baseline_configs.log" ;;
    criterion)
      timeout -k 10 1100 python tools/criterion_grid.py --cpu > "$OUT/criterion_grid.log" 2>&1; rc=$?; tail -14 "$OUT/criterion_grid.log"
      if [ $rc -eq 0 ]; then
        timeout -k 10 1100 python tools/criterion_grid.py --full > "$OUT/criterion_grid_full.log" 2>&1; rc=$?; tail -14 "$OUT/criterion_grid_full.log"
      fi ;;
    rehearsal)
      timeout -k 10 1100 bash tools/rehearsal.sh > "$OUT/rehearsal.log" 2>&1; rc=$?; tail -12 "$OUT/rehearsal.log" ;;
    pytest)
      IFS=: read -r files kexpr <<< "$rest"
      timeout -k 10 1100 python -m pytest ${files//,/ } -m gpu -x -q ${kexpr:+-k "${kexpr//+/ }"} > "$OUT/pytest_some.log" 2>&1; rc=$?; tail -15 "$OUT/pytest_some.log" ;;
    libpytest)
      # libpytest:NAME:FILE[,FILE...][:K+EXPRESSION]   GPU tests against grayscott_amd/variants/libgs_hip_NAME.so
      IFS=: read -r lib files kexpr <<< "$rest"
      GS_HIP_LIBRARY=$ROOT/grayscott_amd/variants/libgs_hip_$lib.so timeout -k 10 1100 python -m pytest ${files//,/ } -m gpu -x -q ${kexpr:+-k "${kexpr//+/ }"} > "$OUT/pytest_$lib.log" 2>&1; rc=$?; tail -8 "$OUT/pytest_$lib.log" ;;
    libpy)
      # libpy:NAME:SCRIPT[:args]   a python tool against grayscott_amd/variants/libgs_hip_NAME.so
      IFS=: read -r lib script pyargs <<< "$rest"
      GS_HIP_LIBRARY=$ROOT/grayscott_amd/variants/libgs_hip_$lib.so timeout -k 10 900 python "$script" ${pyargs//,/ } > "$OUT/$(basename "$script" .py)_$lib.log" 2>&1; rc=$?; tail -12 "$OUT/$(basename "$script" .py)_$lib.log" ;;
    py)
      IFS=: read -r script pyargs <<< "$rest"
      timeout -k 10 900 python "$script" ${pyargs//,/ } > "$OUT/$(basename "$script" .py).log" 2>&1; rc=$?; tail -40 "$OUT/$(basename "$script" .py).log" ;;
    *) echo "unknown stage $stage"; rc=2 ;;
  esac
  echo "=== stage $stage rc=$rc"
  [ $rc -eq 0 ] || exit $rc
done
