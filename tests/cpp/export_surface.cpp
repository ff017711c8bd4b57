// export_surface.cpp — TEST HELPER (tests/test_gpu_interop.py): plays the decoder.  A process of its own allocates a "surface" in device memory
// (hipMemCreate, exportable), fills it with the bytes of <file>, exports the allocation as a POSIX file descriptor and sends that descriptor to the
// test over a unix-domain socket (SCM_RIGHTS) — the way a decoder hands a dma-buf to its consumer.  It then waits for one byte on the socket
// (the consumer is done) and exits.  usage: export_surface <socket path> <file>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <unistd.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "export_surface: %s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char **argv) {
    if (argc < 3) return 1;
    FILE *f = fopen(argv[2], "rb");
    if (!f) return 1;
    std::vector<unsigned char> bytes;
    unsigned char buf[1 << 16];
    for (size_t n; (n = fread(buf, 1, sizeof(buf), f)) > 0;) bytes.insert(bytes.end(), buf, buf + n);
    fclose(f);
    CK(hipSetDevice(0));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
    prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    const size_t size = (bytes.size() + gran - 1) / gran * gran;
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, size, &prop, 0));
    void *ptr = nullptr;
    CK(hipMemAddressReserve(&ptr, size, 0, nullptr, 0));
    CK(hipMemMap(ptr, size, 0, h, 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(ptr, size, &acc, 1));
    CK(hipMemcpy(ptr, bytes.data(), bytes.size(), hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    int fd = -1;
    CK(hipMemExportToShareableHandle(&fd, h, hipMemHandleTypePosixFileDescriptor, 0));
    // hand the descriptor over
    int s = socket(AF_UNIX, SOCK_STREAM, 0);
    sockaddr_un a = {};
    a.sun_family = AF_UNIX; strncpy(a.sun_path, argv[1], sizeof(a.sun_path) - 1);
    if (connect(s, (sockaddr *)&a, sizeof(a)) != 0) { perror("connect"); return 3; }
    unsigned long long sz = size;
    iovec io = {&sz, sizeof(sz)};
    char ctrl[CMSG_SPACE(sizeof(int))] = {};
    msghdr m = {};
    m.msg_iov = &io; m.msg_iovlen = 1; m.msg_control = ctrl; m.msg_controllen = sizeof(ctrl);
    cmsghdr *c = CMSG_FIRSTHDR(&m);
    c->cmsg_level = SOL_SOCKET; c->cmsg_type = SCM_RIGHTS; c->cmsg_len = CMSG_LEN(sizeof(int));
    memcpy(CMSG_DATA(c), &fd, sizeof(int));
    if (sendmsg(s, &m, 0) < 0) { perror("sendmsg"); return 3; }
    char done = 0;
    (void)!read(s, &done, 1);                     // the consumer has released its mapping (or went away)
    close(s); close(fd);
    (void)hipMemUnmap(ptr, size); (void)hipMemAddressFree(ptr, size); (void)hipMemRelease(h);
    return 0;
}
