// <gtsam/nonlinear/ISAM2.h> — STAND-IN (oracle/ref_shim/README.md): see gtsam/lins_ref_gtsam.h
#include <gtsam/lins_ref_gtsam.h>
