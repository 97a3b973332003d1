"""RGB-only social-grouping variant.

Mirror of the public surface of the reference's model/sg_model.py: RGB_Backbone (a duplicate of
the one in gat_model.py, :13-135 -- re-exported here), Tran_SG (:136-211) and
SocialGrouping_model (:215-264), with the same parameter names and shapes.
"""
import torch
import torch.nn as nn

from .gat_model import RGB_Backbone  # noqa: F401  (identical class in the reference)
from ..metric_ops import pairwise_euclidean_distance


def _mlp3(d_in, d_mid, d_out):
    return nn.Sequential(nn.Linear(d_in, d_mid), nn.ReLU(), nn.Linear(d_mid, d_mid), nn.ReLU(), nn.Linear(d_mid, d_out))


class Tran_SG(nn.Module):
    """Group tokens + actor tokens through a Transformer encoder; the pairwise affinity is a
    Gaussian of the Euclidean distance between the phi embeddings (theta, mlp1, mlp_PE and mlp2 are
    constructed but unused by the reference's forward, :191-209)."""

    def __init__(self, d_model=512, nhead=8, N=6, num_token=2, out_feature_dim=256):
        super().__init__()
        self.num_token = num_token
        self.out_feature_dim = out_feature_dim
        self.Group_token = nn.Parameter(torch.randn((num_token, d_model), requires_grad=True))
        self.transformer_encoder = nn.TransformerEncoder(nn.TransformerEncoderLayer(d_model, nhead), num_layers=N)
        self.mlp1 = nn.Sequential(nn.Linear((num_token + 1) * d_model, out_feature_dim), nn.ReLU())
        self.mlp_PE = nn.Sequential(nn.Linear(out_feature_dim + 4, out_feature_dim), nn.Tanh())
        self.mlp2 = nn.Sequential(nn.Linear(2 * out_feature_dim, out_feature_dim), nn.ReLU(),
                                  nn.Linear(out_feature_dim, 1), nn.Sigmoid())
        self.phi = _mlp3((num_token + 1) * d_model + 4, d_model, out_feature_dim)
        self.theta = _mlp3((num_token + 1) * d_model + 4, d_model, out_feature_dim)

    def gaussian_similarity(self, x, sigma=10.0):
        return torch.exp(-torch.pow(pairwise_euclidean_distance(x), 2) / (2 * sigma ** 2))

    def forward(self, F, bboxes):
        out = self.transformer_encoder(torch.concat([self.Group_token, F], dim=0))   # (Ng + N, d_model)
        tokens = out[:self.num_token, :].flatten().unsqueeze(0)
        feats = out[self.num_token:, :]
        joint = torch.concat([tokens.repeat([feats.shape[0], 1]), feats, bboxes], dim=1)
        return self.gaussian_similarity(self.phi(joint))


class SocialGrouping_model(nn.Module):
    def __init__(self, cfg, d_model=512, nhead=8, N=6, num_token=2, out_feature_dim=256):
        super().__init__()
        self.RGB_backbone = RGB_Backbone(cfg.RGB_BACKBONE)
        self.SG_tran = Tran_SG(d_model, nhead, N, num_token, out_feature_dim)
        self.W = 1280
        self.H = 720

    def box_normalizing(self, bboxes):
        bboxes[:, (0, 2)] /= self.W   # in place, as the reference
        bboxes[:, (1, 3)] /= self.H
        return bboxes

    def forward(self, batch):
        (images, bboxes, pcs, bboxes3d, bboxes_num, person_id, social_group_id, seq_id, frame_id, action,
         social_group_activity, data_dict) = batch
        _B, _T, _C, _H, _W = images.shape
        images = images.view(_B, _C, _T, _H, _W)
        rgb = self.RGB_backbone(images, [bboxes[i, :, :] for i in range(bboxes.shape[0])], person_id)
        rgb = rgb.reshape(rgb.shape[0], -1)
        A_theta = self.SG_tran(rgb, self.box_normalizing(bboxes[0, :rgb.shape[0], :]))
        if not self.training:
            A_theta = A_theta.fill_diagonal_(1.)
        return A_theta
