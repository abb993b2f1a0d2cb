A=1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22
for t in 16 14 16 14 16 14 15 15; do
  a=$(grep nr_throttled /sys/fs/cgroup/cpu.stat | cut -d' ' -f2); u=$(grep throttled_usec /sys/fs/cgroup/cpu.stat | cut -d' ' -f2)
  python tools/e2e_bench.py --chroms $A --repeat 4 --threads $t > gpurun_out/r4_h6.log 2>&1 || exit 1
  b=$(grep nr_throttled /sys/fs/cgroup/cpu.stat | cut -d' ' -f2); v=$(grep throttled_usec /sys/fs/cgroup/cpu.stat | cut -d' ' -f2)
  echo "$t $(tail -1 gpurun_out/r4_h6.log | grep -o 'all_seconds[^]]*\]') throttled $((b-a)) periods $(((v-u)/1000)) ms"
done
uptime
