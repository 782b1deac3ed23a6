"""Checkpoint compatibility with the reference's files (host-side).

  * `inflate_positional_embeds`  -- load a T_a-frame checkpoint into a T_b-frame model: temporal embeddings are cut or
    interpolated (run/test_egtea.py:46-96; used at run/test_epic.py:118,152 / run/test_egtea.py:115,145 for both
    `visual.temporal_embed` and the decoder's `temporal_embed`).
  * `load_backbone_checkpoint`   -- LaViLa checkpoint {'state_dict': {'module.<key>': ...}} -> CLIP (run/train.py:433-439).
  * `load_decoder_checkpoint` / `save_runtime_checkpoint` -- the dict the reference writes
    {'epoch','state_dict','best_acc','optimizer','iteration'} with a rolling window of 10 files
    (run/train.py:232-240,523-545; utils/train_utils.py:192-205).
The real checkpoints cannot be downloaded here; the functions are exercised on synthetic state dicts.
"""
import glob
import os
from collections import OrderedDict
from datetime import datetime

import torch
import torch.nn.functional as F


def inflate_positional_embeds(current_model_state_dict, new_state_dict, num_frames=4, load_temporal_fix='bilinear',
                              name='visual.temporal_embed', dim=1):
    """Return `new_state_dict` with `name` ([1, T_load, C]) adapted to `num_frames`: more frames loaded -> cut,
    fewer -> 'zeros' padding or 'interp' (nearest) / 'bilinear' interpolation along time."""
    if name in new_state_dict and name in current_model_state_dict:
        emb = new_state_dict[name]
        t_load, c = emb.shape[dim], emb.shape[-1]
        if t_load > num_frames:
            new_state_dict[name] = emb[:, :num_frames, :]
        elif t_load < num_frames:
            if load_temporal_fix == 'zeros':
                out = torch.zeros([emb.shape[0], num_frames, c], dtype=emb.dtype)
                out[:, :t_load] = emb
            elif load_temporal_fix in ('interp', 'bilinear'):
                mode = 'bilinear' if load_temporal_fix == 'bilinear' else 'nearest'
                out = F.interpolate(emb.unsqueeze(0), (num_frames, c), mode=mode).squeeze(0)
            else:
                raise NotImplementedError(load_temporal_fix)
            new_state_dict[name] = out
        if new_state_dict[name].shape[dim] != current_model_state_dict[name].shape[dim]:
            raise NotImplementedError('Loading models with different spatial resolution / patch number not yet implemented, sorry.')
    return new_state_dict


def strip_module_prefix(state_dict):
    """'module.xxx' -> 'xxx' (checkpoints saved from DistributedDataParallel wrappers, run/train.py:435-437)."""
    return OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in state_dict.items())


def load_backbone_checkpoint(backbone, checkpoint, num_frames=None, strict=True):
    """checkpoint: path or loaded dict with 'state_dict'.  Temporal embeddings are inflated to the model's frame count."""
    if isinstance(checkpoint, (str, os.PathLike)):
        checkpoint = torch.load(checkpoint, map_location='cpu')
    sd = strip_module_prefix(checkpoint['state_dict'] if 'state_dict' in checkpoint else checkpoint)
    cur = backbone.state_dict()
    sd = inflate_positional_embeds(cur, sd, num_frames=num_frames or backbone.visual.num_frames, name='visual.temporal_embed')
    res = backbone.load_state_dict(sd, strict=False)
    bad = [k for k in res.missing_keys if not k.endswith('attn_mask')]
    if strict and (bad or res.unexpected_keys):
        raise RuntimeError(f'backbone checkpoint mismatch: missing={bad} unexpected={res.unexpected_keys}')
    return res


def load_decoder_checkpoint(decoder, checkpoint, optimizer_state=None, num_frames=None):
    """Loads 'state_dict' (decoder weights; temporal / frame-index embeddings inflated) and returns the bookkeeping fields."""
    if isinstance(checkpoint, (str, os.PathLike)):
        checkpoint = torch.load(checkpoint, map_location='cpu')
    sd = strip_module_prefix(checkpoint['state_dict'])
    cur = decoder.state_dict()
    T = num_frames or decoder.num_frames
    sd = inflate_positional_embeds(cur, sd, num_frames=T, name='temporal_embed')
    if 'frame_index.weight' in sd and sd['frame_index.weight'].shape[0] != T:      # [T, C] table: same rule, dim 0
        w = sd['frame_index.weight'][None]
        w = inflate_positional_embeds({'x': cur['frame_index.weight'][None]}, {'x': w}, num_frames=T, name='x')['x']
        sd['frame_index.weight'] = w[0]
    decoder.load_state_dict(sd, strict=True)
    return {k: checkpoint.get(k) for k in ('epoch', 'best_acc', 'iteration', 'optimizer')}


def save_runtime_checkpoint(state, filename, rm_history=True, keep=10):
    """utils/train_utils.py:192-205: timestamped file next to `filename`, keep the newest `keep`."""
    assert filename.endswith('.pth.tar')
    stamp = datetime.now().strftime("%Y_%m_%d_%H_%M")           # the reference's stamp format (minute resolution)
    path = filename.replace('.pth.tar', f'_{stamp}.pth.tar')
    torch.save(state, path)
    if rm_history:
        history = sorted(glob.glob(filename.replace('.pth.tar', '_*.pth.tar')))
        for h in history[:-keep]:
            try:
                os.remove(h)
            except OSError:
                pass
    return path


def make_save_dict(decoder, epoch, best_acc, iteration, optimizer_state):
    """The dict run/train.py:232-237 saves (decoder weights only -- the frozen backbone is never checkpointed).
    `optimizer_state` = TrainStep.state_dict() (torch.optim.AdamW format) or a TrainStep, whose state is taken."""
    if hasattr(optimizer_state, "state_dict") and not isinstance(optimizer_state, dict):
        optimizer_state = optimizer_state.state_dict()
    return {'epoch': epoch, 'state_dict': decoder.state_dict(), 'best_acc': best_acc, 'optimizer': optimizer_state,
            'iteration': iteration}


def resume_train_step(train_step, checkpoint, num_frames=None):
    """run/train.py:523-545: load decoder weights AND optimizer state (AdamW moments, per-parameter step counts, dropout seed
    stream) of a runtime checkpoint into a step.TrainStep.  Returns the bookkeeping fields (epoch, best_acc, iteration)."""
    if isinstance(checkpoint, (str, os.PathLike)):
        checkpoint = torch.load(checkpoint, map_location='cpu', weights_only=False)
    info = load_decoder_checkpoint(train_step.decoder, checkpoint, num_frames=num_frames)
    if info.get('optimizer') is not None:
        train_step.load_state_dict(info['optimizer'])
    if info.get('iteration') is not None:
        train_step.iteration = int(info['iteration'])
    return {k: info[k] for k in ('epoch', 'best_acc', 'iteration')}
