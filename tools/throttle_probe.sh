# is the CLI throttled by the container's CPU quota while it decodes?  (cpu.stat around one run, by the number of reader threads)
python tools/genome_bam_probe.py 0.125 0 > /dev/null
cd /dev/shm/bdx_genome
for t in 16 12 8 24; do
  sleep 2.5
  a=$(grep -E "nr_throttled|throttled_usec|usage_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' ')
  s=$(date +%s.%N)
  BDX_READ_THREADS=$t BDX_TIMING=1 BDX_FOREGROUND=1 /root/repo/bin/breakdancer-max genome_0.125.cfg 2> /tmp/err.txt > /dev/null
  e=$(date +%s.%N)
  b=$(grep -E "nr_throttled|throttled_usec|usage_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' ')
  echo "threads $t wall $(python -c "print('%.3f' % ($e - $s))")"; echo "  before: $a"; echo "  after:  $b"
  grep -E "steady|device decode" /tmp/err.txt | cut -c1-300
done
