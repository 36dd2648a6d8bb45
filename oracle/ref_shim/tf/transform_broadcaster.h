// <tf/transform_broadcaster.h> — STAND-IN (oracle/ref_shim/README.md): sendTransform does nothing.
#ifndef LINS_REF_SHIM_TF_BROADCASTER_
#define LINS_REF_SHIM_TF_BROADCASTER_
#include <tf/transform_datatypes.h>
namespace tf {
struct TransformBroadcaster {
  void sendTransform(const StampedTransform&) {}
};
}  // namespace tf
#endif
