"""Checkpoint ingest with the reference's file format (oai_analysis/segmentation/utils.py:10-45)."""
import os

import torch


def initialize_model(model, optimizer=None, ckpoint_path=None):
    """Strict-load ``checkpoint['model_state_dict']``; returns (finished_epoch, best_score)."""
    finished_epoch, best_score = 0, 0
    if not ckpoint_path:
        raise ValueError("a checkpoint is required for inference (ckpoint_path is empty)")
    if not os.path.isfile(ckpoint_path):
        raise ValueError("=> no checkpoint found at '{}'".format(ckpoint_path))      # utils.py:41
    print("=> loading checkpoint '{}'".format(ckpoint_path))
    checkpoint = torch.load(ckpoint_path, map_location="cpu")
    for key in ("best_score", "reg_best_score", "seg_best_score"):
        if key in checkpoint:
            best_score = checkpoint[key]
            break
    model.load_state_dict(checkpoint["model_state_dict"], strict=True)
    finished_epoch += checkpoint.get("epoch", 0)
    print("=> loaded checkpoint '{}' (epoch {})".format(ckpoint_path, checkpoint.get("epoch", 0)))
    return finished_epoch, best_score
