// <pcl/filters/filter.h> — STAND-IN (oracle/ref_shim/README.md): nothing of this header is used on the compiled path.
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
