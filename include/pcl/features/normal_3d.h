/* forwarding header: see symmicp_pcl_shim.h (stands in for the PCL header of the same name, reference ICP/stdafx.h:8-14) */
#include "symmicp_pcl_shim.h"
