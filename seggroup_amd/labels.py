"""Annotation-derived label files: drop-ins for the label producers of the reference's `seggroup/dataset/scannet/util.py`
(`load_labels` 76-92, `read_label_mapper` 103-110, `load_aggregation` 113-126, `generate_real_labels` 129-170,
`generate_seg_adjacency_matrix` 224-244, `group_adjacency_segs` 252-265, `generate_weak_labels` 268-427,
`generate_real_label_pth` 697-729, `generate_weak_label_pth` 733-768).

They turn ScanNet's annotation files (`<scene>.aggregation.json`, `scannetv2-labels.combined.tsv`, the manual click files) into
the hot path's inputs `label/seg/<style>/resampled/<s>/<s>.label.pth` (weak labels) and `label/real/raw/<s>/<s>.label.pth`
(ground truth).  This is host-side bookkeeping over a few thousand segments -- the per-vertex parts are NumPy gathers, the text
files go through the native formatter (`sg_write_label_txt`) -- with the reference's file names, return values, random draws
(`np.random.randint`, same call order) and quirks (scene0217_00's aggregation is cut at objectId 31; a weak-label cluster other
than an instance's largest is annotated only from 100 points up).  `root` replaces the reference's CWD (`dataset/scannet`).
"""
import csv
import json
import os
from typing import Dict, List, Optional

import numpy as np

from . import hip


def _scene_name(scene_path: str) -> str:
    return os.path.split(scene_path[:-1] if scene_path.endswith("/") else scene_path)[-1]


def _write_txt(path: str, values) -> None:
    os.makedirs(os.path.dirname(path), exist_ok=True)
    v = np.ascontiguousarray(np.asarray(values), dtype=np.int32)
    hip.check(hip.lib().sg_write_label_txt(path.encode(), v.ctypes.data, v.shape[0]))


def load_labels(label_path):
    """util.py:76-92: one integer per line (.txt) or the `segIndices` list (.json) -> list of ints"""
    if label_path.endswith('.txt'):
        with open(label_path, 'rb') as f:
            return np.array(f.read().split(), dtype=np.int64).tolist()
    if label_path.endswith('.json'):
        with open(label_path, 'r') as f:
            return json.load(f)["segIndices"]
    print('Not supported file type!')
    raise SystemExit(1)


def load_seg_labels(label_file):
    with open(label_file, 'r') as f:
        return json.load(f)["segIndices"]


def read_label_mapper(filename, label_from='raw_category', label_to='nyu40id') -> Dict[str, int]:
    assert os.path.isfile(filename)
    with open(filename) as f:
        return {row[label_from]: int(row[label_to]) for row in csv.DictReader(f, delimiter='\t')}


def load_aggregation(aggregation_file, mapper):
    """util.py:113-126 -> (segment -> objectId + 1, segment -> mapped class); later groups overwrite earlier ones"""
    with open(aggregation_file, 'r') as f:
        groups = json.load(f)["segGroups"]
    cut = aggregation_file.split('/')[-1][:12] == 'scene0217_00'
    seg2ins, seg2sem = {}, {}
    for g in groups:
        if cut and g['objectId'] == 31:                        # util.py:120-121
            break
        for s in g['segments']:
            seg2ins[s] = g['objectId'] + 1
            seg2sem[s] = mapper[g['label']]
    return seg2ins, seg2sem


def generate_real_labels(scene_path, root: str = "."):
    """util.py:129-170: label/real/raw/<s>/<s>.{ins,sem}.txt -- per vertex, 0 = unlabeled"""
    scene_path = scene_path[:-1] if scene_path.endswith('/') else scene_path
    name = _scene_name(scene_path)
    seg = np.asarray(load_seg_labels(os.path.join(scene_path, name + '_vh_clean_2.0.010000.segs.json')), dtype=np.int64)
    mapper = read_label_mapper(os.path.join('/'.join(scene_path.split('/')[:-2]), 'scannetv2-labels.combined.tsv'))
    seg2ins, seg2sem = load_aggregation(os.path.join(scene_path, name + '.aggregation.json'), mapper)
    uniq, inv = np.unique(seg, return_inverse=True)
    ins = np.array([seg2ins.get(int(s), 0) for s in uniq], dtype=np.int64)[inv]
    sem = np.array([seg2sem.get(int(s), 0) for s in uniq], dtype=np.int64)[inv]
    out = os.path.join(root, 'label', 'real', 'raw', name)
    _write_txt(os.path.join(out, name + '.ins.txt'), ins)
    _write_txt(os.path.join(out, name + '.sem.txt'), sem)


def generate_seg_adjacency_matrix(mesh_path, seg_path, plydata=None):
    """util.py:224-244: [S,S] 0/1 matrix, 1 where a mesh edge joins two different segments"""
    from .prepare import mesh_arrays, read_ply
    seg = np.asarray(load_labels(seg_path), dtype=np.int64)
    if plydata is None:
        plydata = read_ply(mesh_path)
    _, _, faces = mesh_arrays(plydata)
    f = np.asarray(faces, dtype=np.int64).reshape(-1, 3)
    n = int(seg.max()) + 1
    adj = np.zeros([n, n])
    for a, b in ((0, 1), (0, 2), (1, 2)):
        s1, s2 = seg[f[:, a]], seg[f[:, b]]
        m = s1 != s2
        adj[s1[m], s2[m]] = 1
        adj[s2[m], s1[m]] = 1
    return adj


def group_adjacency_segs(adjacency_matrix, segs) -> List[List[int]]:
    """util.py:252-265: connected groups of `segs` -- in the reference's list order (the cluster of the LATER segment absorbs the
    earlier one's and the absorbed list is removed), which the callers' argmax / random picks depend on"""
    clusters = [[int(s)] for s in segs]
    where = {int(s): clusters[i] for i, s in enumerate(segs)}           # segment -> the list object that holds it
    for i in range(len(segs)):
        for j in range(i):
            a, b = int(segs[i]), int(segs[j])
            if adjacency_matrix[a, b] == 0:
                continue
            ca, cb = where[a], where[b]
            if ca is cb:
                continue
            ca.extend(cb)
            for s in cb:
                where[s] = ca
            clusters.pop(next(k for k, c in enumerate(clusters) if c is cb))
    return clusters


def generate_weak_labels(scene_path, plydata=None, label_style='manual', manual_label_path=None, main_num=-1, anno_num=1, root: str = "."):
    """util.py:268-427: label/seg/<style>/raw/<s>/<s>.{ins,sem}.txt (-1 = no click on the vertex's segment) ->
    (labeled vertices, vertices, clicked segments, segments, instances)"""
    scene_path = scene_path[:-1] if scene_path.endswith('/') else scene_path
    name = _scene_name(scene_path)
    raw = os.path.join(root, 'label', 'real', 'raw', name)
    ins_labels = np.array(load_labels(os.path.join(raw, name + '.ins.txt')))
    sem_labels = np.array(load_labels(os.path.join(raw, name + '.sem.txt')))
    ins_unique = np.unique(ins_labels)
    picked: List[int] = []

    if label_style == 'manual':
        seg_labels = np.array(load_seg_labels(os.path.join(scene_path, name + '_vh_clean_2.0.010000.segs.json')))
        with open(os.path.join(manual_label_path, name + '.json'), 'r') as f:
            manual = json.load(f)
        for ins in manual:
            picked.extend(int(s) for s in manual[ins])
    else:
        seg_path = os.path.join(raw, name + '.seg.txt')
        seg_labels = np.array(load_labels(seg_path))
        adjacency = generate_seg_adjacency_matrix(os.path.join(scene_path, name + '_vh_clean_2.ply'), seg_path, plydata)
        seg_points = np.bincount(seg_labels, minlength=int(seg_labels.max()) + 1)

        def annotate(ids, counts):
            """one cluster's clicks: `ids` its segments by descending size (cut to main_num), `counts` their sizes"""
            if label_style == 'maxseg':
                picked.extend(ids[:anno_num].tolist())
            elif label_style == 'rand':
                picked.append(ids[np.random.randint(low=0, high=len(ids))])
            elif label_style == 'mainseg':
                for i in range(anno_num):
                    if i >= len(ids):
                        continue
                    while True:                                # size-weighted draw among the main segments, without repeats
                        r = np.random.randint(low=0, high=np.sum(counts))
                        index = 0
                        for index in range(main_num):
                            if r < np.sum(counts[:index + 1]):
                                break
                        if ids[index] not in picked:
                            picked.append(ids[index])
                            break

        for ins in ins_unique:
            if ins == 0:
                continue                                       # unlabeled vertices
            ins_segs = np.unique(seg_labels[ins_labels == ins])
            clusters = group_adjacency_segs(adjacency, ins_segs)
            sizes, mains, main_counts = [], [], []
            for c in clusters:
                n = seg_points[np.asarray(c)]
                sizes.append(int(n.sum()))
                order = np.argsort(-n)                         # the reference's (unstable) argsort of the negated sizes
                top = order if main_num == -1 else order[:main_num]
                mains.append(np.asarray(c)[top])
                main_counts.append((-np.sort(-n)) if main_num == -1 else (-np.sort(-n))[:main_num])
            best = int(np.argmax(sizes))
            annotate(mains[best], main_counts[best])
            for j in range(len(clusters)):
                if j == best or sizes[j] < 100:                # util.py:356-359
                    continue
                annotate(mains[j], main_counts[j])

    seg_unique = np.unique(seg_labels)
    hit = np.isin(seg_labels, np.asarray(picked, dtype=np.int64)) if picked else np.zeros(seg_labels.shape, bool)
    ins_weak = np.where(hit, ins_labels, -1)
    sem_weak = np.where(hit, sem_labels, -1)
    style = label_style + ('_' + str(main_num) if label_style == 'mainseg' else '') + ('_a' + str(anno_num) if anno_num > 1 else '')
    out = os.path.join(root, 'label', 'seg', style, 'raw', name)
    _write_txt(os.path.join(out, name + '.ins.txt'), ins_weak)
    _write_txt(os.path.join(out, name + '.sem.txt'), sem_weak)
    ins_num = ins_unique.shape[0] - (1 if ins_unique[0] == 0 else 0)
    return int(np.count_nonzero(ins_weak != -1)), int(ins_weak.shape[0]), len(picked), int(seg_unique.shape[0]), int(ins_num)


def generate_real_label_pth(scene_path, root: str = "."):
    """util.py:697-729: label/real/raw/<s>/<s>.label.pth = LongTensor [V,2] (sem, ins), 0 = unlabeled"""
    import torch
    name = _scene_name(scene_path)
    raw = os.path.join(root, 'label', 'real', 'raw', name)
    ins = np.array(load_labels(os.path.join(raw, name + '.ins.txt')), dtype=np.int64)
    sem = np.array(load_labels(os.path.join(raw, name + '.sem.txt')), dtype=np.int64)
    torch.save(torch.from_numpy(np.stack([sem, ins], axis=1)), os.path.join(raw, name + '.label.pth'))


def generate_weak_label_pth(scene_name, label_style='manual', root: str = "."):
    """util.py:733-768: label/seg/<style>/resampled/<s>/<s>.label.pth = LongTensor [num_points,2] (sem, ins) of the SAMPLED points,
    classes / instances counted from 0, -1 = no weak label"""
    import torch
    raw = os.path.join(root, 'label', 'seg', label_style, 'raw', scene_name)
    ins = np.array(load_labels(os.path.join(raw, scene_name + '.ins.txt')), dtype=np.int64)
    sem = np.array(load_labels(os.path.join(raw, scene_name + '.sem.txt')), dtype=np.int64)
    labeled = ins >= 0                                         # util.py:741-743: both columns shift where the INSTANCE label is >= 0
    ins = np.where(labeled, ins - 1, ins)
    sem = np.where(labeled, sem - 1, sem)
    mapper = torch.load(os.path.join(root, 'data', 'resampled', scene_name, scene_name + '.map.pth'))
    out = os.path.join(root, 'label', 'seg', label_style, 'resampled', scene_name)
    os.makedirs(out, exist_ok=True)
    torch.save(torch.from_numpy(np.stack([sem, ins], axis=1))[mapper], os.path.join(out, scene_name + '.label.pth'))
