"""Tiny detector forward+backward on the GPU vs the oracle (filled in as the conv stack lands)."""


def run():
    pass
