// <nav_msgs/Odometry.h> — STAND-IN (oracle/ref_shim/README.md): the message's fields.
#ifndef LINS_REF_SHIM_NAV_MSGS_ODOMETRY_
#define LINS_REF_SHIM_NAV_MSGS_ODOMETRY_
#include <boost/shared_ptr.hpp>
#include <geometry_msgs/Quaternion.h>
#include <std_msgs/Header.h>

#include <string>
namespace nav_msgs {
struct Odometry {
  typedef boost::shared_ptr<Odometry> Ptr;
  typedef boost::shared_ptr<const Odometry> ConstPtr;
  std_msgs::Header header;
  std::string child_frame_id;
  geometry_msgs::PoseWithCovariance pose;
  geometry_msgs::TwistWithCovariance twist;
};
}  // namespace nav_msgs
#endif
