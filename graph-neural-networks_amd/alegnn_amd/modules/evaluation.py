"""evaluate / evaluateSingleNode -- the reference's alegnn/modules/evaluation.py:18-168.

Test-set cost of the 'Best' and the 'Last' checkpoint; returns {'costBest', 'costLast'} and (doSaveVars) pickles it to
<saveDir>/evalVars/<name>evalVars.pkl.  Under torch.distributed every rank evaluates its identical replica; rank 0 writes.
"""
from __future__ import annotations

import os
import pickle

import torch
import torch.distributed as dist


def _run(model, data, forward, kwargs):
    doSaveVars = kwargs.get('doSaveVars', True)
    xTest, yTest = data.getSamples('test')
    xTest = xTest.to(model.device)
    yTest = yTest.to(model.device)
    evalVars = {}
    for label in ('Best', 'Last'):                                  # evaluation.py:58-71
        model.load(label=label)
        with torch.no_grad():
            evalVars['cost' + label] = float(data.evaluate(forward(xTest), yTest))
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    if doSaveVars and rank == 0:
        saveDirVars = os.path.join(model.saveDir, 'evalVars')
        os.makedirs(saveDirVars, exist_ok=True)
        with open(os.path.join(saveDirVars, model.name + 'evalVars.pkl'), 'wb') as f:
            pickle.dump(evalVars, f)
    return evalVars


def evaluate(model, data, **kwargs):
    return _run(model, data, lambda x: model.archit(x), kwargs)


def evaluateSingleNode(model, data, **kwargs):
    assert 'singleNodeForward' in dir(model.archit)                 # evaluation.py:109-110
    assert 'getLabelID' in dir(data)
    targetIDs = data.getLabelID('test')
    return _run(model, data, lambda x: model.archit.singleNodeForward(x, targetIDs), kwargs)
