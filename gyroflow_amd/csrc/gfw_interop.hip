// gfw_interop.hip — decoder / interop surfaces as warp sources and destinations without the host.
//
// The reference's zero-copy path imports memory another API owns and maps it to a device pointer of the compute API
// (src/core/gpu/wgpu_interop_cuda.rs:181-215: cuImportExternalMemory -> cuExternalMemoryGetMappedBuffer; the plane descriptors travel as
// BufferSource::CUDABuffer, src/core/gpu/mod.rs:67-70, src/rendering/zero_copy.rs:67-112).  The MI355X equivalent: a POSIX file descriptor that
// names a device allocation — a dma-buf exported by the video decoder (VA-API / VCN), by Vulkan (vkGetMemoryFdKHR), or by another process through
// hipMemExportToShareableHandle — is imported here and comes back as a plain device pointer for a GFW_BUF_HIP_DEVICE buffer description.
//
// Only LINEAR (pitch-linear) layouts can be warped in place: the kernels address pixels as base + y * stride + x * bytes.  The caller states the
// surface's DRM format modifier; anything but DRM_FORMAT_MOD_LINEAR (0) is refused — a tiled decoder surface must be exported linear (VA-API:
// VA_EXPORT_SURFACE_SEPARATE_LAYERS with a linear target, or a VPP copy) before it gets here.  Pitch is not a property of the import: it is the
// `stride` of the gfw_buffer_desc the caller builds on top of the pointer (plane offsets are added by the caller the same way).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/gfwarp.h"

void gfw_set_error_text(const char *text);        // gfw_api.hip

struct gfw_external {
    int kind = 0;                                  // 1 hipImportExternalMemory, 2 virtual-memory import (hipMemImportFromShareableHandle + map)
    hipExternalMemory_t ext = nullptr;
    hipMemGenericAllocationHandle_t vmm = {};
    void *ptr = nullptr;
    size_t size = 0, mapped = 0;
    int device = 0;
};

static void err(const char *what, hipError_t e) {
    char buf[384];
    snprintf(buf, sizeof(buf), "%s failed: %s", what, hipGetErrorString(e));
    gfw_set_error_text(buf);
}

extern "C" int gfw_import_external_fd(int fd, size_t size, unsigned long long drm_format_modifier, void **dev_ptr_out, gfw_external **handle_out) {
    if (fd < 0 || size == 0 || !dev_ptr_out || !handle_out) { gfw_set_error_text("gfw_import_external_fd: bad arguments"); return GFW_ERR_INVALID_ARGUMENT; }
    *dev_ptr_out = nullptr; *handle_out = nullptr;
    if (drm_format_modifier != 0ull) {             // DRM_FORMAT_MOD_LINEAR
        char buf[160];
        snprintf(buf, sizeof(buf), "surface layout modifier 0x%llx is not linear: export the surface pitch-linear", drm_format_modifier);
        gfw_set_error_text(buf);
        return GFW_ERR_UNSUPPORTED_BUFFER;
    }
    {   // The HIP runtime does not survive a descriptor that is not a dma-buf (a regular file crashed hipImportExternalMemory on the GPU box): look first.
        // A dma-buf is an anonymous inode whose /proc link reads "/dmabuf:<name>" or "anon_inode:dmabuf".
        char link[64], target[256];
        snprintf(link, sizeof(link), "/proc/self/fd/%d", fd);
        const ssize_t n = readlink(link, target, sizeof(target) - 1);
        if (n <= 0) { gfw_set_error_text("gfw_import_external_fd: not an open file descriptor"); return GFW_ERR_INVALID_ARGUMENT; }
        target[n] = 0;
        // (a prefix match on the link text — "/dmabuf:" or "anon_inode:dmabuf" — plus fstat: a regular file, directory, device node or socket whose PATH merely
        // contains the word is not one; a dma-buf's inode is none of those types)
        struct stat st;
        const bool named = strncmp(target, "/dmabuf:", 8) == 0 || strncmp(target, "anon_inode:dmabuf", 17) == 0 || strncmp(target, "anon_inode:[dmabuf", 18) == 0;
        const bool plain = fstat(fd, &st) != 0 || S_ISREG(st.st_mode) || S_ISDIR(st.st_mode) || S_ISCHR(st.st_mode) || S_ISBLK(st.st_mode) || S_ISSOCK(st.st_mode) || S_ISFIFO(st.st_mode);
        if (!named || plain) {
            char buf[384];
            snprintf(buf, sizeof(buf), "descriptor %d (%s) is not a dma-buf: nothing the device can map", fd, target);
            gfw_set_error_text(buf);
            return GFW_ERR_UNSUPPORTED_BUFFER;
        }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { gfw_set_error_text("no HIP device visible"); return GFW_ERR_NO_DEVICE; }
    gfw_external *h = new gfw_external();
    h->size = size;
    (void)hipGetDevice(&h->device);
    // (1) the external-memory interface, what the reference's CUDA interop uses.  The runtime takes ownership of a file descriptor it imports
    // successfully (CUDA's contract), so it gets a duplicate: the caller's fd stays the caller's.
    {
        const int dupfd = dup(fd);
        if (dupfd >= 0) {
            hipExternalMemoryHandleDesc d = {};
            d.type = hipExternalMemoryHandleTypeOpaqueFd; d.handle.fd = dupfd; d.size = size; d.flags = 0;
            hipError_t e = hipImportExternalMemory(&h->ext, &d);
            if (e == hipSuccess) {
                hipExternalMemoryBufferDesc b = {};
                b.offset = 0; b.size = size; b.flags = 0;
                e = hipExternalMemoryGetMappedBuffer(&h->ptr, h->ext, &b);
                if (e == hipSuccess && h->ptr) { h->kind = 1; *dev_ptr_out = h->ptr; *handle_out = h; return GFW_OK; }
                (void)hipDestroyExternalMemory(h->ext); h->ext = nullptr;
            } else {
                close(dupfd);
            }
            (void)hipGetLastError();
        }
    }
    // (2) a shareable handle of the virtual-memory interface (hipMemCreate + hipMemExportToShareableHandle in the exporting process)
    {
        hipError_t e = hipMemImportFromShareableHandle(&h->vmm, (void *)(uintptr_t)fd, hipMemHandleTypePosixFileDescriptor);
        if (e != hipSuccess) { err("hipImportExternalMemory / hipMemImportFromShareableHandle", e); delete h; return GFW_ERR_UNSUPPORTED_BUFFER; }
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = h->device;
        size_t gran = 0;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || gran == 0) gran = 2u << 20;
        h->mapped = (size + gran - 1) / gran * gran;
        e = hipMemAddressReserve(&h->ptr, h->mapped, 0, nullptr, 0);
        if (e == hipSuccess) {
            e = hipMemMap(h->ptr, h->mapped, 0, h->vmm, 0);
            if (e == hipSuccess) {
                hipMemAccessDesc acc = {};
                acc.location.type = hipMemLocationTypeDevice; acc.location.id = h->device; acc.flags = hipMemAccessFlagsProtReadWrite;
                e = hipMemSetAccess(h->ptr, h->mapped, &acc, 1);
                if (e == hipSuccess) { h->kind = 2; *dev_ptr_out = h->ptr; *handle_out = h; return GFW_OK; }
                (void)hipMemUnmap(h->ptr, h->mapped);
            }
            (void)hipMemAddressFree(h->ptr, h->mapped);
        }
        (void)hipMemRelease(h->vmm);
        err("mapping the imported allocation", e);
        delete h;
        return GFW_ERR_HIP;
    }
}

extern "C" int gfw_release_external(gfw_external *h) {
    if (!h) return GFW_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();                  // kernels that still read or write the mapping
    if (h->kind == 1) {
        if (h->ptr) (void)hipFree(h->ptr);
        if (h->ext) (void)hipDestroyExternalMemory(h->ext);
    } else if (h->kind == 2) {
        (void)hipMemUnmap(h->ptr, h->mapped);
        (void)hipMemAddressFree(h->ptr, h->mapped);
        (void)hipMemRelease(h->vmm);
    }
    delete h;
    return GFW_OK;
}
