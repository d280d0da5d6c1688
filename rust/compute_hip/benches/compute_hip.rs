// Same one-line harness as the other backends (e.g. compute/block/benches/compute_block.rs).
// `cpu_benchmark!` = the "compute" (perform_steps) and "full" (perform_steps + make_result_view)
// workloads of compute/shared/src/benchmark.rs:77-93.  `gpu_benchmark!` cannot be used: its third
// workload is written against the Vulkan-only SimulateGpu trait and ImageConcentration
// (benchmark.rs:97-113, behind compute's "gpu" feature).
compute::cpu_benchmark!(compute_hip);
