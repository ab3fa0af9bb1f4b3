"""Dataloader of the ScanNet dataset: drop-in for the reference's `seggroup/data.py:18-41`.

`ScanNet(label_style)[i] -> (data[N,6] f32, weak_label[N,2] i64, info[1] i64)` read from the reference's
CWD-relative tree (`dataset/scannet/...`); `root` lets a caller point at another directory.
"""
import os

import torch
from torch.utils.data import Dataset


class ScanNet(Dataset):
    def __init__(self, label_style='manual', root='.'):
        self.label_style = label_style
        self.data_root = os.path.join(root, 'dataset', 'scannet')
        with open(os.path.join(self.data_root, 'scannetv2_train.txt'), 'r') as f:
            self.scene_list = f.readlines()

    def __getitem__(self, item):
        scene_name = self.scene_list[item][:-1]
        base = os.path.join(self.data_root, 'data', 'resampled', scene_name, scene_name)
        data = torch.load(base + '.pcl.pth')
        weak_label = torch.load(os.path.join(self.data_root, 'label', 'seg', self.label_style, 'resampled', scene_name,
                                             scene_name + '.label.pth'))
        info = torch.load(base + '.info.pth')
        return data, weak_label, info

    def __len__(self):
        return len(self.scene_list)
