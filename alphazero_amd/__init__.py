"""MI355X-native batched MCTS engine behind the reference's `cmcts.mcts` API.

    alphazero_amd.engine   ctypes binding of libazmcts.so (the C ABI in include/azmcts.h)
    alphazero_amd.cmcts    drop-in for the reference's `cmcts` module (class `mcts`)
    alphazero_amd.selfplay batched self-play / arena drivers (reference src/self_play.py, src/eval.py)

The search always runs on the GPU through libazmcts.so; importing the engine
without the built library raises (there is no CPU fallback).
"""
__all__ = ["engine", "cmcts"]
