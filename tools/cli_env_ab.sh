# interleaved A/B of one environment switch of the CLI on the genome-share BAM: tools/cli_env_ab.sh VAR=VALUE [pairs]
python tools/genome_bam_probe.py 0.125 0 > /dev/null
cd /dev/shm/bdx_genome
for i in $(seq 1 ${2:-6}); do
  for v in "" "$1"; do
    sleep 2.5
    s=$(date +%s.%N)
    env $v BDX_FOREGROUND=1 /root/repo/bin/breakdancer-max genome_0.125.cfg > /dev/null 2>&1
    e=$(date +%s.%N)
    echo "${v:-default} $(python -c "print('%.3f' % ($e - $s))")"
  done
done | sort | awk '{a[$1]=a[$1]" "$2} END {for (k in a) print k":"a[k]}'
