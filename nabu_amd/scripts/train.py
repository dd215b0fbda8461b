"""Training entry point (reference: nabu/scripts/train.py:13-112): reads
database.conf, model.cfg, trainer.cfg and validation_evaluator.cfg from the
experiment directory, builds the trainer and trains.  One process per GPU; launch
several with ``python -m torch.distributed.run --nproc-per-node N -m
nabu_amd.scripts.train --expdir DIR`` for the data-parallel mode."""
import argparse
import os
from configparser import ConfigParser

from nabu_amd.computing import dist
from nabu_amd.neuralnetworks.trainers import trainer_factory


def train(clusterfile, job_name, task_index, ssh_command, expdir, testing=False):
    '''does everything for asr training (same signature as the reference;
    clusterfile / job_name / ssh_command belong to the parameter-server launch of
    the reference and are ignored: the process group comes from the environment)'''
    def read(name):
        parser = ConfigParser()
        parser.read(os.path.join(expdir, name))
        return parser
    database_cfg = read('database.conf')
    model_cfg = read('model.cfg')
    trainer_cfg = read('trainer.cfg')
    evaluator_cfg = read('validation_evaluator.cfg')
    server = dist.create_server()
    tr = trainer_factory.factory(trainer_cfg.get('trainer', 'trainer'))(
        conf=trainer_cfg, dataconf=database_cfg, modelconf=model_cfg, evaluatorconf=evaluator_cfg,
        expdir=expdir, server=server, task_index=server.rank if task_index is None else task_index)
    print('starting training')
    return tr.train(testing)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--clusterfile', default=None)
    ap.add_argument('--job_name', default='local')
    ap.add_argument('--task_index', type=int, default=None)
    ap.add_argument('--ssh_command', default='None')
    ap.add_argument('--expdir', default='expdir')
    a = ap.parse_args()
    train(a.clusterfile, a.job_name, a.task_index, a.ssh_command, a.expdir)
