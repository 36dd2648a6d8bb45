// <tf/transform_datatypes.h> — STAND-IN (oracle/ref_shim/README.md): nothing of this header is used on the compiled path.
