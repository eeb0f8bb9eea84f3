# Builds the MI355X (gfx950) Gibbs library and the CPU parity oracle.
#   make            -> grample_amd/libgrample_hip.so + oracle/liboracle.so
# hipcc cross-compiles for gfx950 without a GPU present.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := grample_amd/csrc
OBJDIR := build/obj
# -ffp-contract=off: the conditional-threshold arithmetic is shared by the
# sweep kernel, the memoisation kernel and the probe; it must round the same
# way in all three.
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-result
LIB := grample_amd/libgrample_hip.so
OBJS := $(OBJDIR)/kernels.o $(OBJDIR)/kernels_1.o $(OBJDIR)/kernels_2.o $(OBJDIR)/sweep_bin.o $(OBJDIR)/sweep_par4_1.o $(OBJDIR)/sweep_par4_2.o $(OBJDIR)/sweep_par4_3.o $(OBJDIR)/sweep_par4_4.o $(OBJDIR)/sweep_par4_5.o $(OBJDIR)/sweep_pw4.o $(OBJDIR)/sweep_dir4.o $(OBJDIR)/sweep_chroma.o $(OBJDIR)/sweep_bits.o $(OBJDIR)/sampler.o $(OBJDIR)/host_model.o $(OBJDIR)/compiled_model.o

TOOL := tools/grample_hip_sample

all: $(LIB) oracle $(TOOL) ubench

# stand-alone micro-benchmarks behind bench.py's roofline and DESIGN.md (issue rates; byte state vs bit-packed state)
UBENCH := scripts/ubench/issue_rate.bin scripts/ubench/state_form.bin
ubench: $(UBENCH)
scripts/ubench/%.bin: scripts/ubench/%.hip
	$(HIPCC) --offload-arch=$(ARCH) -O2 -o $@ $<

# C++ host driver above the C ABI (links nothing but the library)
$(TOOL): tools/grample_hip_sample.cpp include/grample_hip.h $(LIB)
	g++ -O2 -std=c++17 -Wall -o $@ $< -Lgrample_amd -lgrample_hip -Wl,-rpath,'$$ORIGIN/../grample_amd' -Wl,-rpath,/opt/rocm/lib

$(OBJDIR)/kernels.o: $(CSRC)/kernels.hip $(CSRC)/kernels.h $(CSRC)/device_layout.h $(CSRC)/philox.h $(CSRC)/wave_prio.h $(CSRC)/cond.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

# the instantiations of the generic k_sweep for Card <= 16 and Card <= 64 (kernels.hip, GR_KSWEEP_PART)
$(OBJDIR)/kernels_%.o: $(CSRC)/kernels.hip $(CSRC)/kernels.h $(CSRC)/device_layout.h $(CSRC)/philox.h $(CSRC)/wave_prio.h $(CSRC)/cond.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -DGR_KSWEEP_PART=$* -c $< -o $@

$(OBJDIR)/sweep_bin.o: $(CSRC)/sweep_bin.hip $(CSRC)/kernels.h $(CSRC)/device_layout.h $(CSRC)/philox.h $(CSRC)/wave_prio.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

# the instantiations of k_sweep_par4, by record shape (sweep_bin.hip, GR_PAR4_PART): four translation units built side by side
$(OBJDIR)/sweep_par4_%.o: $(CSRC)/sweep_bin.hip $(CSRC)/kernels.h $(CSRC)/device_layout.h $(CSRC)/philox.h $(CSRC)/wave_prio.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -DGR_PAR4_PART=$* -c $< -o $@

$(OBJDIR)/sweep_chroma.o: $(CSRC)/sweep_chroma.hip $(CSRC)/kernels.h $(CSRC)/device_layout.h $(CSRC)/philox.h $(CSRC)/wave_prio.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OBJDIR)/sweep_bits.o: $(CSRC)/sweep_bits.hip $(CSRC)/kernels.h $(CSRC)/device_layout.h $(CSRC)/philox.h $(CSRC)/wave_prio.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OBJDIR)/sweep_dir4.o: $(CSRC)/sweep_dir4.hip $(CSRC)/kernels.h $(CSRC)/device_layout.h $(CSRC)/philox.h $(CSRC)/wave_prio.h $(CSRC)/cond.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

# max-ilp: the value-parallel kernel runs two independent conditionals per wavefront (pw4_cond<2>); the default scheduler
# emits them one after the other, this one interleaves their dependent f64 chains
$(OBJDIR)/sweep_pw4.o: $(CSRC)/sweep_pw4.hip $(CSRC)/kernels.h $(CSRC)/device_layout.h $(CSRC)/philox.h $(CSRC)/wave_prio.h $(CSRC)/cond.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -mllvm -amdgpu-sched-strategy=max-ilp -c $< -o $@

$(OBJDIR)/%.o: $(CSRC)/%.cpp $(wildcard $(CSRC)/*.h) $(wildcard $(CSRC)/*.hpp) include/grample_hip.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -ldl -Wl,-rpath,/opt/rocm/lib

oracle:
	$(MAKE) -s -C oracle liboracle.so

clean:
	rm -rf build $(LIB) $(TOOL)
	$(MAKE) -s -C oracle clean

.PHONY: all oracle clean ubench
