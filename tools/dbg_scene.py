import sys, os, json, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import make_fixture_scene, load_golden
from test_gpu_scene import _run
from seggroup_amd import weights
from oracle import cpu_ref as O
name = sys.argv[1] if len(sys.argv) > 1 else 'tiny_4k'
idx = json.load(open(os.path.join(ROOT, 'tests/golden/index.json')))
W = weights.load_npz(os.path.join(ROOT, 'tests/golden/weights_g2.npz'))
sc = make_fixture_scene(idx, name)
res, t, pipe = _run(sc, W, 'ins_infer', debug=True)
ref = O.forward_scene(sc, W, 'ins_infer', keep=True)
print('trace gpu', res.trace, 'oracle', ref['trace'])
for i, nm in enumerate(('mlp_2', 'mlp_3')):
    members = t['members'][i].cpu().numpy()
    pf = np.empty((sc.num_points, 64), np.float32); pf[members] = t['pf'][i].cpu().numpy()
    st = ref['stages'][nm]
    err = np.abs(pf - st['point_feat'])
    knn_pos = t['knn'][i].cpu().numpy()
    knn_pts = np.empty((sc.num_points, 20), np.int64); knn_pts[members] = members[knn_pos]
    bad_knn = np.any(np.sort(knn_pts, 1) != np.sort(st['knn'], 1), axis=1)
    print(nm, 'pf max err', err.max(), 'rows>1e-4', int((err.max(1) > 1e-4).sum()), 'knn rows differ', int(bad_knn.sum()),
          'pf err on knn-equal rows', err[~bad_knn].max())
    ordiff = np.nonzero(np.any(knn_pts != st['knn'], axis=1))[0]
    print('   rows with order/any diff', len(ordiff))
    for q in ordiff[:3]:
        xyz = sc.data[:, :3]
        print('   q', q, 'gpu', knn_pts[q], '\n        ref', st['knn'][q])
        print('      gpu scores', O.knn_scores(xyz[q][None], xyz[knn_pts[q]])[0])
        print('      ref scores', O.knn_scores(xyz[q][None], xyz[st['knn'][q]])[0])
    # cluster sizes
    root = st['root']
    if bad_knn.any():
        r = np.nonzero(bad_knn)[0][:5]
        for q in r:
            csize = int((root == root[q]).sum())
            print('   row', q, 'cluster size', csize, 'gpu', np.sort(knn_pts[q])[:8], 'ref', np.sort(st['knn'][q])[:8])
print('stage ms', {k: round(v, 3) for k, v in pipe.stage_times().items()})
