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
// image_projection_node.cpp (the node in front of StateEstimator) is a class around a NodeHandle: it subscribes to the
// raw cloud and publishes its results.  Here a subscription does nothing (the _ref driver calls the handler itself) and
// a publication stores a copy of the message under its topic, where the driver picks it up.
#include <boost/shared_ptr.hpp>
#include <std_msgs/Header.h>

#include <map>
#include <string>
namespace lins_ref_shim {
template <class M>
struct Published {
  static std::map<std::string, M>& by_topic() {
    static thread_local std::map<std::string, M> m;
    return m;
  }
};
}  // namespace lins_ref_shim
namespace ros {
class Subscriber {};
class Publisher {
 public:
  Publisher() {}
  explicit Publisher(const std::string& topic) : topic_(topic) {}
  template <class M>
  void publish(const M& msg) const {
    ::lins_ref_shim::Published<M>::by_topic()[topic_] = msg;
  }
  unsigned getNumSubscribers() const { return 0; }

 private:
  std::string topic_;
};
class NodeHandle {
 public:
  NodeHandle() {}
  explicit NodeHandle(const std::string&) {}
  template <class M, class T>
  Subscriber subscribe(const std::string&, unsigned, void (T::*)(const boost::shared_ptr<const M>&), T*) {
    return Subscriber();
  }
  template <class M>
  Publisher advertise(const std::string& topic, unsigned, bool = false) {
    return Publisher(topic);
  }
};
inline void init(int&, char**, const std::string&) {}
inline void spin() {}
inline void spinOnce() {}
inline bool ok() { return false; }
struct Rate {
  explicit Rate(double) {}
  void sleep() {}
};
}  // namespace ros
#endif
