// lins_ref_shim/events.h — thread-local observation points shared by the stand-in headers (oracle/ref_shim):
// the number of kd-tree queries issued so far and the ROS log lines emitted so far (each stamped with that
// count).  The _ref driver reads them to recover what the reference keeps in locals; the reference's own text is
// not touched.
#ifndef LINS_REF_SHIM_EVENTS_
#define LINS_REF_SHIM_EVENTS_
#include <string>
#include <vector>
namespace lins_ref_shim {
struct Event {
  std::string text;
  long queries;
};
inline long& kdtree_queries() {
  static thread_local long n = 0;
  return n;
}
inline std::vector<Event>& events() {
  static thread_local std::vector<Event> log;
  return log;
}
inline void reset_events() {
  kdtree_queries() = 0;
  events().clear();
}
inline void ros_event(const std::string& text) {
  Event e;
  e.text = text;
  e.queries = kdtree_queries();
  events().push_back(e);
}
inline void ros_event_fmt(const char* fmt, ...) { ros_event(std::string(fmt)); }
}  // namespace lins_ref_shim
#endif
