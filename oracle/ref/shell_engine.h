// shell_engine.h — TEST INFRASTRUCTURE. An Engine that owns nothing but an allocator, for driving the reference's own World
// (engine/world.cpp compiled in place by oracle/Makefile) outside the engine: World's constructor asks the Engine for the allocator and
// the - empty - system list; World::serialize asks it to compress (restated from engine/engine.cpp:254-269 on the vendored LZ4).
// Shared by oracle/ref/world_shim.cpp (the ref_world_* C entry points) and tests/cpp/real_header_harness.cpp. Mine, not the reference's.
#pragma once

#include <stdlib.h>
#include <string.h>

#include "core/allocator.h"
#include "core/array.h"
#include "core/delegate_list.h"
#include "core/page_allocator.h"
#include "core/path.h"
#include "core/stream.h"
#include "core/string.h"
#include "engine/engine.h"
#include "engine/plugin.h"
#include "engine/world.h"

extern "C" int LZ4_compress_fast(const char* src, char* dst, int srcSize, int dstCapacity, int acceleration);
extern "C" int LZ4_decompress_safe(const char* src, char* dst, int compressedSize, int dstCapacity);
extern "C" int LZ4_compressBound(int inputSize);

namespace lmx_ref {
using namespace Lumix;

struct HeapAllocator final : IAllocator {
	void* allocate(size_t size, size_t align) override {
		void* p = nullptr;
		if (posix_memalign(&p, align < sizeof(void*) ? sizeof(void*) : align, size ? size : 1) != 0) abort();
		return p;
	}
	void deallocate(void* ptr) override { free(ptr); }
	void* reallocate(void* ptr, size_t new_size, size_t old_size, size_t align) override {
		if (new_size == 0) { free(ptr); return nullptr; }
		void* p = allocate(new_size, align);
		if (ptr) { memcpy(p, ptr, old_size < new_size ? old_size : new_size); free(ptr); }
		return p;
	}
};

struct NoSystems final : SystemManager {
	NoSystems(IAllocator& a) : systems(a), libraries(a), loaded(a) {}
	void initSystems() override {}
	void unload(ISystem*) override {}
	ISystem* load(const char*) override { return nullptr; }
	void addSystem(ISystem*, void*) override {}
	void update(float) override {}
	ISystem* getSystem(const char*) override { return nullptr; }
	const Array<ISystem*>& getSystems() const override { return systems; }
	const Array<void*>& getLibraries() const override { return libraries; }
	void* getLibrary(ISystem*) const override { return nullptr; }
	DelegateList<void(void*)>& libraryLoaded() override { return loaded; }
	Array<ISystem*> systems;
	Array<void*> libraries;
	DelegateList<void(void*)> loaded;
};

[[noreturn]] inline void unused() { abort(); }

struct ShellEngine final : Engine {
	ShellEngine() : systems(heap) {}
	void init() override {}
	World& createWorld() override { unused(); }
	void destroyWorld(World&) override {}
	void setMainWindow(os::WindowHandle) override {}
	os::WindowHandle getMainWindow() override { return os::WindowHandle(); }
	FileSystem& getFileSystem() override { unused(); }
	InputSystem& getInputSystem() override { unused(); }
	SystemManager& getSystemManager() override { return systems; }
	ResourceManagerHub& getResourceManager() override { unused(); }
	PageAllocator& getPageAllocator() override { unused(); }
	IAllocator& getAllocator() override { return heap; }
	EntityPtr instantiatePrefab(World&, const PrefabResource&, const DVec3&, const Quat&, const Vec3&, EntityMap&) override { unused(); }
	void startGame(World&) override {}
	void stopGame(World&) override {}
	void update(World&) override {}
	DeserializeProjectResult deserializeProject(InputMemoryStream&, Path&) override { unused(); }
	void serializeProject(OutputMemoryStream&, const Path&) const override {}
	float getLastTimeDelta() const override { return 0; }
	void setTimeMultiplier(float) override {}
	void pause(bool) override {}
	bool isPaused() const override { return false; }
	void nextFrame() override {}
	bool decompress(Span<const u8> src, Span<u8> dst) override { // engine/engine.cpp:254-258
		const int res = LZ4_decompress_safe((const char*)src.begin(), (char*)dst.begin(), (int)src.length(), (int)dst.length());
		return res == (int)dst.length();
	}
	bool compress(Span<const u8> src, OutputMemoryStream& dst) override { // engine/engine.cpp:260-269
		const int cap = LZ4_compressBound((int)src.length());
		const u64 start = dst.size();
		dst.resize(start + cap);
		// LZ4_compress_fast_extState(state, ..., acceleration 1) in the reference; LZ4_compress_fast is the same call on a local state
		const int res = LZ4_compress_fast((const char*)src.begin(), (char*)dst.getMutableData() + start, (int)src.length(), cap, 1);
		if (res == 0) return false;
		dst.resize(start + res);
		return true;
	}
	HeapAllocator heap;
	NoSystems systems;
};

} // namespace lmx_ref
