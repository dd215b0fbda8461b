"""Decoding entry point (reference: nabu/scripts/decode.py:12-72): decodes the data sets named in
<expdir>/recognizer.cfg with the trained model and writes the results to <expdir>/decoded."""
import argparse
import os
from configparser import ConfigParser

from nabu_amd.neuralnetworks.recognizer import Recognizer
from nabu_amd.scripts.test import load_model


def decode(expdir, testing=False):
    '''does everything for decoding'''
    database_cfg, recognizer_cfg = ConfigParser(), ConfigParser()
    database_cfg.read(os.path.join(expdir, 'database.conf'))
    recognizer_cfg.read(os.path.join(expdir, 'recognizer.cfg'))
    model = load_model(expdir, testing)
    recognizer = Recognizer(model=model, conf=recognizer_cfg, dataconf=database_cfg, expdir=expdir)
    if testing:
        return None
    return recognizer.recognize()


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--expdir', default='expdir')
    decode(ap.parse_args().expdir, False)
