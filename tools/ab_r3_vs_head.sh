cd /root/repo
for rep in 1 2; do
 for S in 192 48; do
  echo -n "HEAD rep $rep: "; python tools/train_bench.py --samples $S --steps 240 --warmup 24 --ray-batch random 2>/dev/null | tail -1
  echo -n "r3   rep $rep: "; (cd _r3 && python tools/train_bench.py --samples $S --steps 240 --warmup 24 --ray-batch random 2>/dev/null | tail -1)
 done
done
python tools/train_regress_probe.py 2>&1 | grep -v Warning
