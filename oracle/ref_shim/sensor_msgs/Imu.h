// <sensor_msgs/Imu.h> — STAND-IN (oracle/ref_shim/README.md): the message's fields.
#ifndef LINS_REF_SHIM_SENSOR_MSGS_IMU_
#define LINS_REF_SHIM_SENSOR_MSGS_IMU_
#include <boost/shared_ptr.hpp>
#include <geometry_msgs/Quaternion.h>
#include <std_msgs/Header.h>
namespace sensor_msgs {
struct Imu {
  typedef boost::shared_ptr<Imu> Ptr;
  typedef boost::shared_ptr<const Imu> ConstPtr;
  std_msgs::Header header;
  geometry_msgs::Quaternion orientation;
  geometry_msgs::Vector3 angular_velocity, linear_acceleration;
};
}  // namespace sensor_msgs
#endif
