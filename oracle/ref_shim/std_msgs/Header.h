// <std_msgs/Header.h> — STAND-IN (oracle/ref_shim/README.md): the three fields of the message.
#ifndef LINS_REF_SHIM_STD_MSGS_HEADER_
#define LINS_REF_SHIM_STD_MSGS_HEADER_
#include <cstdint>
#include <string>
namespace ros {
struct Time {
  double sec_;
  Time() : sec_(0.0) {}
  explicit Time(double s) : sec_(s) {}
  double toSec() const { return sec_; }
  Time& fromSec(double s) {
    sec_ = s;
    return *this;
  }
  static Time now() { return Time(); }
};
}  // namespace ros
namespace std_msgs {
struct Header {
  std::uint32_t seq;
  ros::Time stamp;
  std::string frame_id;
  Header() : seq(0) {}
};
}  // namespace std_msgs
#endif
