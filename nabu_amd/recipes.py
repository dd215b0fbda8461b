"""Recipe helpers: read the reference-format .cfg files of a recipe directory."""
import os
from configparser import ConfigParser

RECIPES = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'config', 'recipes')


def read_cfg(path):
    parser = ConfigParser()
    if not parser.read(path):
        raise Exception('cannot read %s' % path)
    return parser


def from_dict(sections):
    """ConfigParser from {section: {key: value}} (all values stringified)."""
    parser = ConfigParser()
    for s, kv in sections.items():
        parser.add_section(s)
        for k, v in kv.items():
            parser.set(s, k, str(v))
    return parser


def load_recipe(recipe, **overrides):
    """(modelconf, trainerconf, evaluatorconf) of config/recipes/<recipe> (or a path).
    overrides: 'section.key' -> value, e.g. **{'encoder.num_units': 32}."""
    d = recipe if os.path.isdir(recipe) else os.path.join(RECIPES, recipe)
    confs = [read_cfg(os.path.join(d, f)) for f in ('model.cfg', 'trainer.cfg')]
    ev = os.path.join(d, 'validation_evaluator.cfg')
    confs.append(read_cfg(ev) if os.path.exists(ev) else from_dict({'evaluator': {'evaluator': 'None'}}))
    for key, value in overrides.items():
        section, field = key.split('.', 1)
        for c in confs:
            if c.has_section(section):
                c.set(section, field, str(value))
    return tuple(confs)
