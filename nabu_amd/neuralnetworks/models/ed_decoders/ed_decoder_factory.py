"""EDDecoder factory (reference: models/ed_decoders/ed_decoder_factory.py:5-24)."""


def factory(decoder):
    '''get an EDDecoder class by its recipe name'''
    if decoder == 'speller':
        from nabu_amd.neuralnetworks.models.ed_decoders import speller
        return speller.Speller
    elif decoder == 'dnn_decoder':
        from nabu_amd.neuralnetworks.models.ed_decoders import dnn_decoder
        return dnn_decoder.DNNDecoder
    elif decoder == 'hotstart_decoder':
        raise Exception('decoder type hotstart_decoder is outside the MI355X hot path')
    else:
        raise Exception('undefined decoder type: %s' % decoder)
