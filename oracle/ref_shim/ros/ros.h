// <ros/ros.h> — STAND-IN (oracle/ref_shim/README.md).
//
// The reference reports filter divergence only through ROS_WARN (StateEstimator.hpp:560, 567, 586) and the ICP
// fallback's convergence only through ROS_INFO_STREAM (SE:1189); the stand-in macros append the text to a
// thread-local event log (lins_ref_shim/events.h) that the _ref driver reads back, so those branches are
// observable without touching the reference's text.
#ifndef LINS_REF_SHIM_ROS_
#define LINS_REF_SHIM_ROS_
#include <lins_ref_shim/events.h>

#include <sstream>
#define ROS_WARN(...) ::lins_ref_shim::ros_event_fmt(__VA_ARGS__)
#define ROS_INFO(...) ::lins_ref_shim::ros_event_fmt(__VA_ARGS__)
#define ROS_ERROR(...) ::lins_ref_shim::ros_event_fmt(__VA_ARGS__)
#define LINS_REF_SHIM_STREAM(x)              \
  do {                                       \
    std::ostringstream lins_ref_shim_ss;     \
    lins_ref_shim_ss << x;                   \
    ::lins_ref_shim::ros_event(lins_ref_shim_ss.str()); \
  } while (0)
#define ROS_WARN_STREAM(x) LINS_REF_SHIM_STREAM(x)
#define ROS_INFO_STREAM(x) LINS_REF_SHIM_STREAM(x)
#define ROS_ERROR_STREAM(x) LINS_REF_SHIM_STREAM(x)
namespace ros {
class NodeHandle;
}
#endif
