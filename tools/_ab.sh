cd $GRAFT_REPO_ROOT
python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('value',d['value'],'kernel_ms',d['roofline']['kernel_ms'],'frac',d['roofline']['frac'],'prep',d['roofline']['prepare_ms_per_call'],'split',d.get('split_precision',{}).get('value'),'train',d.get('train_step',{}).get('ms_per_step'), d.get('train_step',{}).get('fwd_ms'), d.get('train_step',{}).get('bwd_ms'))"
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
