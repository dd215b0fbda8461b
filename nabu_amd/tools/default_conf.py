"""Per-class default configuration merging — same contract as the reference's
nabu/tools/default_conf.py:9-36: every field of the class' ``defaults/<class>.cfg``
that is missing from ``conf`` is filled in; a default with an EMPTY value marks a
required field and raises."""
import os
from configparser import ConfigParser


def apply_defaults(conf, default_file):
    """conf: dict of strings read from the recipe; default_file: path of the
    ``[default]`` cfg.  Returns the updated conf (also modified in place)."""
    if os.path.exists(default_file):
        parser = ConfigParser()
        parser.read(default_file)
        for field, value in parser.items('default'):
            if field not in conf:
                if value == '':
                    raise Exception(
                        'the field %s was not found in the configuration file' % field)
                conf[field] = value
    return conf


def defaults_path(module_file, obj):
    """<dir of module>/defaults/<classname lowercased>.cfg (ed_encoder.py:28-33)."""
    return os.path.join(os.path.dirname(os.path.realpath(module_file)), 'defaults',
                        type(obj).__name__.lower() + '.cfg')
