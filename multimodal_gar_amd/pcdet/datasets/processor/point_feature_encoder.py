"""Mirror of the reference's pcdet/datasets/processor/point_feature_encoder.py:4-57 (``PointFeatureEncoder``): picks the
configured feature columns of a point array.  Host-side, numpy (or torch tensors: only slicing / concatenation is used)."""
import numpy as np
import torch


class PointFeatureEncoder(object):
    def __init__(self, config, point_cloud_range=None):
        self.point_encoding_config = config
        assert list(config.src_feature_list[0:3]) == ['x', 'y', 'z']
        self.used_feature_list = config.used_feature_list
        self.src_feature_list = config.src_feature_list
        self.point_cloud_range = point_cloud_range

    @property
    def num_point_features(self):
        return getattr(self, self.point_encoding_config.encoding_type)(points=None)

    def forward(self, data_dict):
        """data_dict['points'] (N, 3 + C_in) -> (N, 3 + C_out); sets data_dict['use_lead_xyz']."""
        data_dict['points'], data_dict['use_lead_xyz'] = getattr(self, self.point_encoding_config.encoding_type)(data_dict['points'])
        if self.point_encoding_config.get('filter_sweeps', False) and 'timestamp' in self.src_feature_list:
            idx = self.src_feature_list.index('timestamp')
            dt = np.round(np.asarray(data_dict['points'][:, idx]), 2)
            steps = sorted(np.unique(dt))
            max_dt = steps[min(len(steps) - 1, self.point_encoding_config.max_sweeps - 1)]
            data_dict['points'] = data_dict['points'][dt <= max_dt]
        return data_dict

    def absolute_coordinates_encoding(self, points=None):
        if points is None:
            return len(self.used_feature_list)
        assert points.shape[-1] == len(self.src_feature_list)
        cols = [points[:, 0:3]]
        for name in self.used_feature_list:
            if name in ('x', 'y', 'z'):
                continue
            idx = self.src_feature_list.index(name)
            cols.append(points[:, idx:idx + 1])
        cat = torch.cat if torch.is_tensor(points) else np.concatenate
        return cat(cols, 1), True
