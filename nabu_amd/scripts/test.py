"""Test entry point (reference: nabu/scripts/test.py:16-110): evaluates the trained model of an
experiment directory with the evaluator of test_evaluator.cfg (normally a decoder_evaluator:
beam-search decoding + label error rate), prints the loss and writes it to <expdir>/result."""
import argparse
import os
import pickle
from configparser import ConfigParser

import numpy as np

from nabu_amd.neuralnetworks.evaluators import evaluator_factory
from nabu_amd.neuralnetworks.models.model import Model


def _read(expdir, name):
    parser = ConfigParser()
    parser.read(os.path.join(expdir, name))
    return parser


def load_model(expdir, testing=False):
    '''the model of an experiment: its configuration from model/model.pkl (model.cfg when
    testing), its variables from model/network.ckpt.npz when that exists'''
    trainer_cfg = _read(expdir, 'trainer.cfg')
    pkl = os.path.join(expdir, 'model', 'model.pkl')
    if testing or not os.path.exists(pkl):
        model_cfg = _read(expdir, 'model.cfg')
    else:
        with open(pkl, 'rb') as fid:
            model_cfg = ConfigParser()
            model_cfg.read_dict(pickle.load(fid))
    model = Model(conf=model_cfg, trainlabels=int(trainer_cfg.get('trainer', 'trainlabels')), constraint=None)
    ckpt = os.path.join(expdir, 'model', 'network.ckpt.npz')
    if not testing and os.path.exists(ckpt):
        with np.load(ckpt) as state:
            model.store.restore_from({k: state[k] for k in state.files})
    return model


def test(expdir, testing=False):
    '''does everything for testing'''
    database_cfg = _read(expdir, 'database.conf')
    model = load_model(expdir, testing)
    evaluator_cfg = _read(expdir, 'test_evaluator.cfg')
    evaluator = evaluator_factory.factory(evaluator_cfg.get('evaluator', 'evaluator'))(
        conf=evaluator_cfg, dataconf=database_cfg, model=model)
    loss, update_loss, numbatches = evaluator.evaluate()
    if testing:
        return None
    for i in range(numbatches):
        update_loss(i)
    print('loss = %f' % loss[0])
    with open(os.path.join(expdir, 'result'), 'w') as fid:
        fid.write(str(loss[0]))
    return loss[0]


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--expdir', default='expdir')
    test(ap.parse_args().expdir, False)
