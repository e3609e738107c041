// hip/hip_ext.h of tests/hostsim (TEST INFRASTRUCTURE, see hip_runtime.h): hipExtLaunchKernelGGL records the two events around the launch
#pragma once
#include <hip/hip_runtime.h>

#define hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, start_event, stop_event, flags, ...) \
	do {                                                                                                 \
		if (start_event) (void)hipEventRecord(start_event, stream);                                      \
		::hostsim::launch(#kernel, kernel, grid, block, lds, stream, __VA_ARGS__);                               \
		if (stop_event) (void)hipEventRecord(stop_event, stream);                                        \
	} while (0)
