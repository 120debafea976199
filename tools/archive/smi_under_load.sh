rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -v "^$" | head -30
echo ---- under load
python bench.py --cpu-num-vars 0 --steps 4000 > /tmp/b.json 2>/dev/null &
BP=$!
sleep 6
for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>&1 | grep -i "sclk\|mclk\|fclk\|power\|socclk" ; echo; sleep 0.7; done
wait $BP
python -c "import json; d=json.loads(open('/tmp/b.json').read()); print(d['ms_per_step'])"
