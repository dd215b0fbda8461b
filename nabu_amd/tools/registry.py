"""String -> class dispatch shared by the recipe factories.

The reference dispatches with one if/elif chain per package (trainer_factory, ed_encoder_factory,
ed_decoder_factory, evaluator_factory, tfreader_factory, tfwriter_factory); here every factory is
a table {recipe name: 'module:Class'} resolved lazily, plus the names that exist in the reference
but lie outside the MI355X hot path (they raise with an explanation instead of 'undefined')."""
import importlib


class Registry(object):
    def __init__(self, kind, table, outside=(), undefined='undefined %s type: %s'):
        self.kind, self.table, self.outside, self.undefined = kind, dict(table), tuple(outside), undefined

    def __call__(self, name):
        target = self.table.get(name)
        if target is None:
            if name in self.outside:
                raise Exception('%s type %s exists in the reference but is outside the MI355X hot path '
                                '(SURVEY.md section 8)' % (self.kind, name))
            raise Exception(self.undefined % (self.kind, name))
        module, cls = target.split(':')
        return getattr(importlib.import_module(module), cls)

    def names(self):
        return sorted(self.table)
