# Builds the gfx950 HIP library in-tree (the .so travels to the GPU box with the snapshot).
#   make            the product library: -fvisibility=hidden, only the UFV_API entry points of include/ufv.h are dynamic symbols
#   make EXTRA=-DUFV_LAB_KERNELS   lab build: the diagnostic attention kernel ids too (tools/lab/build_variant_lib.sh)
#   make asan       host-side sanitizer build + run of the library's host logic (tests/asan/: cost model, split-K ring, error paths) -- CPU only
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := ufvideo_amd/csrc
SRCS  := $(CSRC)/gemm.hip $(CSRC)/gemm256.hip $(CSRC)/gemm256_b.hip $(CSRC)/gemm256_q.hip $(CSRC)/gemm256_s.hip $(CSRC)/gemm256_r.hip $(CSRC)/gemm256_m.hip $(CSRC)/gemm256_m2.hip $(CSRC)/gemm_state.hip $(CSRC)/attn.hip $(CSRC)/ops.hip $(CSRC)/qwen2_decode.hip $(CSRC)/sam_heads.hip $(CSRC)/quant.hip $(CSRC)/loss.hip $(CSRC)/resize.hip $(CSRC)/train.hip $(CSRC)/sample.hip $(CSRC)/train_proj.hip $(CSRC)/attn_bwd.hip $(CSRC)/seg_train.hip $(CSRC)/stages.hip
OBJS  := $(SRCS:.hip=.o)
DEPS  := $(SRCS:.hip=.d)
LIB   := ufvideo_amd/libufv_hip.so
EXTRA ?=
FLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -fvisibility=hidden -Wall -Wno-unused-function -Iinclude $(EXTRA)

all: $(LIB)

# header dependencies are the compiler's own (-MMD): a generated attention body rebuilds attn.o only
$(CSRC)/%.o: $(CSRC)/%.hip Makefile
	$(HIPCC) $(FLAGS) -MMD -MP -MF $(CSRC)/$*.d -c $< -o $@

$(LIB): $(OBJS) $(CSRC)/ufv_exports.map
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -fvisibility=hidden -Wl,--version-script=$(CSRC)/ufv_exports.map -o $@ $(OBJS)

-include $(DEPS)

# ---- `make asan`: the library's HOST code under AddressSanitizer + UBSan, on the CPU.  Every source is compiled host-only (kernel bodies become launch stubs),
# linked against tests/asan/hip_stub.cpp instead of libamdhip64, and tests/asan/host_logic.cpp drives the cost model, ufv_gemm's dispatch, the split-K flag ring
# (wrap-around, two threads), the error word and the no-device path.  The __hip_fatbin_* symbols the host objects refer to (the embedded device code of a normal
# build) are defined as empty arrays.
ASAN_DIR   := build/asan
ASAN_FLAGS := --cuda-host-only -O1 -g -fPIC -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -Iinclude -Wno-unused-command-line-argument
ASAN_OBJS  := $(patsubst $(CSRC)/%.hip,$(ASAN_DIR)/%.o,$(SRCS))

$(ASAN_DIR)/%.o: $(CSRC)/%.hip Makefile
	@mkdir -p $(ASAN_DIR)
	$(HIPCC) $(ASAN_FLAGS) -c $< -o $@

$(ASAN_DIR)/host_logic: $(ASAN_OBJS) tests/asan/hip_stub.cpp tests/asan/host_logic.cpp
	nm -u $(ASAN_OBJS) | grep -o '__hip_fatbin_[0-9a-f]*' | sort -u | sed 's/.*/extern "C" { char &[8]; }/' > $(ASAN_DIR)/fatbin_syms.cpp
	$(HIPCC) $(ASAN_FLAGS) -c $(ASAN_DIR)/fatbin_syms.cpp -o $(ASAN_DIR)/fatbin_syms.o
	$(HIPCC) $(ASAN_FLAGS) -c tests/asan/hip_stub.cpp -o $(ASAN_DIR)/hip_stub.o
	$(HIPCC) $(ASAN_FLAGS) -c tests/asan/host_logic.cpp -o $(ASAN_DIR)/host_logic.o
	$(HIPCC) -fsanitize=address,undefined -o $@ $(ASAN_DIR)/host_logic.o $(ASAN_DIR)/hip_stub.o $(ASAN_DIR)/fatbin_syms.o $(ASAN_OBJS) -lpthread

asan: $(ASAN_DIR)/host_logic
	ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 $(ASAN_DIR)/host_logic

clean:
	rm -f $(OBJS) $(DEPS) $(LIB)
	rm -rf $(ASAN_DIR)

.PHONY: all clean asan
