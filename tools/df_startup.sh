# tools/df_startup.sh -- the fixed cost of a DF process: a tiny input, wall time against DF's own accounting
W=/tmp/dfstart; rm -rf $W; mkdir -p $W
cp tests/golden/reads.fastb tests/golden/reads.qualp tests/golden/reads.bci $W/
for i in 1 2 3; do
  t0=$(date +%s.%N)
  superplus_amd/DF ROOT=$W LR=$W/reads.fastb PIPELINE=cs ALIGN=False NUM_THREADS=16 MAX_MEM_GB=640 GRAPH=False > $W/out$i.txt 2> $W/err$i.txt
  t1=$(date +%s.%N)
  echo "wall $(python3 -c "print(round($t1 - $t0, 3))") s; $(grep -o '"total_s": [0-9.]*' $W/out$i.txt)"
done
