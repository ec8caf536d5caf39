"""
Checkpoint I/O of the reference's models/model_utils.py:161-211, the on-disk contract of SURVEY.md §8(b):

    save_checkpoint(state, is_best, model_name, output_path, is_master_proc=True, filename='checkpoint.pth.tar')
    load_checkpoint(model, checkpoint_path, classifier=False, is_master_proc=True) -> (start_epoch, best_prec1)

`state` is the dict the training loop builds (online_train.py:757-761): {'epoch', 'state_dict', 'best_prec1'}.
A state_dict taken from a DistributedDataParallel wrapper carries a `module.` prefix on every key; loading strips
it, so checkpoints move freely between wrapped / un-wrapped models and between the reference's modules and this
package's (same 129 / 72 keys, shapes and dtypes).  Pure host code: no kernel involved.
"""
import os
import shutil
from collections import OrderedDict

import torch

_PREFIX = 'module.'


def _checkpoint_dir(output_path, model_name):
    return os.path.join(output_path, "tnet_checkpoints/%s/" % (model_name))


def save_checkpoint(state, is_best, model_name, output_path, is_master_proc=True, filename='checkpoint.pth.tar'):
    """only the master process writes; `is_best` also copies the file to model_best.pth.tar"""
    if not is_master_proc:
        return
    directory = _checkpoint_dir(output_path, model_name)
    os.makedirs(directory, exist_ok=True)
    path = directory + filename
    torch.save(state, path)
    print('\n=> checkpoint:{} saved...'.format(path))
    if is_best:
        best = os.path.join(directory, 'model_best.pth.tar')
        shutil.copyfile(path, best)
        print('=> best_model saved as:{}'.format(best))


def strip_module_prefix(state_dict, classifier=False):
    """keys without the DDP `module.` prefix (the reference cuts the first 7 characters of any key containing it);
    classifier=True additionally drops the projection head (fc*, bn_proj*) of un-prefixed keys, like the reference"""
    out = OrderedDict()
    for k, v in state_dict.items():
        if _PREFIX in k:
            out[k[len(_PREFIX):]] = v
        elif classifier and ('fc' in k or 'bn_proj' in k):
            continue
        else:
            out[k] = v
    return out


def load_checkpoint(model, checkpoint_path, classifier=False, is_master_proc=True):
    """loads `checkpoint['state_dict']` into `model` (strict unless classifier=True); a missing file is reported and,
    as in the reference, leaves nothing to return — here that is an explicit FileNotFoundError instead of the reference's
    UnboundLocalError"""
    if not os.path.isfile(checkpoint_path):
        if is_master_proc:
            print("=> no checkpoint found at '{}'".format(checkpoint_path))
        raise FileNotFoundError(checkpoint_path)
    if is_master_proc:
        print("=> loading checkpoint '{}'".format(checkpoint_path))
    checkpoint = torch.load(checkpoint_path, map_location='cpu')
    start_epoch, best_prec1 = checkpoint['epoch'], checkpoint['best_prec1']
    model.load_state_dict(strip_module_prefix(checkpoint['state_dict'], classifier), strict=not classifier)
    if is_master_proc:
        print("=> loaded checkpoint '{}' (epoch {})".format(checkpoint_path, start_epoch))
    return start_epoch, best_prec1
