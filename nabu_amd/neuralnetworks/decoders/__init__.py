"""Inference decoders: turn a trained model's outputs into label sequences
(the role of nabu/neuralnetworks/decoders)."""
