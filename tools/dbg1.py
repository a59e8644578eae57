import sys, numpy as np
sys.path.insert(0, '.')
from tests.helpers import oracle_case, device_case
from moshpp_amd import capi
from oracle import stageii_oracle as so
which = sys.argv[1]
if which.startswith('mano'):
    case = oracle_case('mano', F=6, M=24, seed=5); dev = device_case(case, optimize_fingers=which.endswith('_f'))
elif which == 'smplh_noprior':
    case = oracle_case('smplh', F=6, M=53, seed=0); dev = device_case(case)
    m = case['m']
    root, body, finger, st1, st2 = so.pose_id_sets('smplh', m['NP'], False, False)
    dev['opts'] = capi.make_opts(so.stageii_weights_default(), st1, st2, [], [])
    dev['prior'] = None
if which == 'smplh_small':
    case = oracle_case('smplh', F=6, M=53, seed=0); dev = device_case(case)
    dev['opts'] = capi.make_opts(so.stageii_weights_default(), [0,1,2,3,4,5], [0,1,2,3,4,5], [], [])
    dev['prior'] = None
print(which, 'NP', case['m']['NP'], flush=True)
out = capi.chain_solve_host(dev['model'], dev['prior'], dev['opts'], [dict(attach=dev['attach'], obs=case['obs'], vis=case['vis'], first=True)])[0]
print(capi.last_launch_info(), out['iters'][:, 0], out['status'], flush=True)
