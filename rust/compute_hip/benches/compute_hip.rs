compute::cpu_benchmark!(compute_hip);
