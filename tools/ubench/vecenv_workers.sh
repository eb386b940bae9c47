# us per VecEnv::step for 4096 / 1024 mock environments against the number of pool threads (host only; which box type did we get?)
g++ -O2 -std=c++17 -pthread -I ppo_cpp_amd/host -o gpurun_out/vecenv_bench tools/ubench/vecenv_bench.cpp || exit 1
grep -m1 "model name" /proc/cpuinfo; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat
for spin in 0 250; do for gap in 0 100; do for w in 1 2 4 8 16; do echo -n "spin $spin "; PPO_VECENV_SPIN_US=$spin gpurun_out/vecenv_bench ${N:-4096} 200 $w $gap | tail -1; done; done; done
grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat
