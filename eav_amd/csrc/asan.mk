# Host-side sanitizer build of libeav_hip.so - kept out of the main Makefile and out of the GPU-box snapshot
# (.gpurunignore): sanitizer builds are CPU-only on this pool.   make -C eav_amd/csrc -f asan.mk
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
SRCS = $(wildcard *.hip)
OBJS = $(SRCS:.hip=.o)

# Host-side sanitizer build (AddressSanitizer + UBSan on the HOST half of the C ABI: argument validation, plan / size
# functions, launch-parameter arithmetic; device code is compiled unsanitised - GPU ASan is not available on this pool).
# tests/test_abi_sanitizer_cpu.py loads it in a child process (LD_PRELOAD of the ASan runtime) and drives every entry point
# that returns before it would touch a GPU.
ASAN_DIR = asan
ASAN_FLAGS = --offload-arch=$(ARCH) -O1 -g -std=c++17 -fPIC -Wall -Wno-unused-function -fsanitize=address,undefined \
             -fno-gpu-sanitize -fno-omit-frame-pointer
ASAN_OBJS = $(addprefix $(ASAN_DIR)/,$(OBJS))
ASAN_TARGET = ../libeav_hip_asan.so

.PHONY: asan
asan: $(ASAN_TARGET)

$(ASAN_DIR)/%.o: %.hip eav_common.h ../../include/eav_hip.h
	@mkdir -p $(ASAN_DIR)
	$(HIPCC) $(ASAN_FLAGS) -c $< -o $@

$(ASAN_TARGET): $(ASAN_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize $(ASAN_OBJS) -o $@


