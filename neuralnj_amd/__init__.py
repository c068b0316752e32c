"""neuralnj_amd: MI355X-native hot path of NeuralNJ (encoder + neural NJ loop).

Host mirror of the reference's model.py / environment.py / utils.py / phydata.py
surface over a C-ABI HIP library (include/nnj.h).  See DESIGN.md.
"""
__version__ = "0.1.0"
