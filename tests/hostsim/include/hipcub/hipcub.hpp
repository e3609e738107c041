// hipcub/hipcub.hpp of tests/hostsim (TEST INFRASTRUCTURE, see hip/hip_runtime.h): the two device-wide primitives the C ABI calls,
// with the library's two-call protocol (first call with a null temporary buffer returns the size).
#pragma once
#include <hip/hip_runtime.h>

#include <numeric>
#include <vector>

namespace hipcub {

struct DeviceScan {
	template <typename In, typename Out>
	static hipError_t ExclusiveSum(void* temp, size_t& temp_bytes, In in, Out out, int n, hipStream_t = nullptr) {
		if (!temp) { temp_bytes = 256; return hipSuccess; }
		typename std::remove_reference<decltype(out[0])>::type acc = 0;
		for (int i = 0; i < n; ++i) {
			const auto v = in[i];
			out[i] = acc;
			acc += v;
		}
		return hipSuccess;
	}
};

struct DeviceRadixSort {
	template <typename K, typename V>
	static hipError_t SortPairs(void* temp, size_t& temp_bytes, const K* keys_in, K* keys_out, const V* values_in, V* values_out, int n, int begin_bit = 0,
		int end_bit = sizeof(K) * 8, hipStream_t = nullptr) {
		if (!temp) { temp_bytes = 256; return hipSuccess; }
		std::vector<uint32_t> order((size_t)n);
		std::iota(order.begin(), order.end(), 0u);
		const K mask = (end_bit - begin_bit >= (int)sizeof(K) * 8) ? ~K(0) : (((K(1) << (end_bit - begin_bit)) - 1) << begin_bit);
		std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return (keys_in[a] & mask) < (keys_in[b] & mask); });
		for (int i = 0; i < n; ++i) {
			keys_out[i] = keys_in[order[(size_t)i]];
			values_out[i] = values_in[order[(size_t)i]];
		}
		return hipSuccess;
	}
};

} // namespace hipcub
