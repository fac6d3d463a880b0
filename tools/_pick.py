"""stdin: one bench.py JSON line -> `value` and the side figures in M agent-steps/s on one line (GPU-pass shell loops)."""
import json,sys
d=json.loads(sys.stdin.read())
print('value %.1f M'%(d['value']/1e6), ' '.join('%s %.1f'%(k.replace('_side_figure',''), d[k].get('value',0)/1e6) for k in ('stage2_side_figure','fidelity_side_figure','reference_shaped_obs_side_figure','rollout_side_figure','train_side_figure') if k in d))
