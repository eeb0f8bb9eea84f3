// Source of the cgo binding described in INTEGRATION.md section 2.  Not compiled in this
// repository (the image has no Go toolchain); it is the file a grample maintainer adds as
// sampler/hip.go, built with `go build -tags hip`.
//go:build hip

package sampler

/*
#cgo CFLAGS: -I${SRCDIR}/../include
#cgo LDFLAGS: -L${SRCDIR}/../lib -lgrample_hip -Wl,-rpath,${SRCDIR}/../lib
#include <stdlib.h>
#include "grample_hip.h"
*/
import "C"

import (
	"runtime"
	"unsafe"

	"github.com/CraigKelly/grample/model"
	"github.com/pkg/errors"
)

// HIPChains is n chains (+ their FullSampler) living on one MI355X.
type HIPChains struct {
	h        *C.gr_sampler
	base     *model.Model
	nChains  int
	variants int
}

// call runs one C-ABI call and, when it fails, reads gr_last_error() ON THE SAME OS THREAD:
// the message is thread-local in the library and a goroutine may migrate between two cgo calls.
func call(f func() C.int) error {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if f() != 0 {
		return errors.New(C.GoString(C.gr_last_error()))
	}
	return nil
}

// cArray is a C-heap array (C.malloc): the model description handed to gr_model_create may
// not point at Go memory that itself holds Go pointers (cgo pointer rule), so every array of
// the description lives on the C heap for the duration of the call.
type cArray struct{ p unsafe.Pointer }

func newCArray(bytes int) cArray {
	if bytes < 1 {
		bytes = 1 // never hand C a nil array (empty scopes / tables)
	}
	return cArray{C.malloc(C.size_t(bytes))}
}
func (a cArray) free() { C.free(a.p) }

// flatten copies the Go model into a C-heap description and creates the library's model from
// it.  The library copies everything at call time, so the arrays are freed before returning.
func flatten(m *model.Model) (*C.gr_model, error) {
	n, f := len(m.Vars), len(m.Funcs)
	nScope, nTab := 0, 0
	for _, fn := range m.Funcs {
		if fn.IsLog {
			return nil, errors.Errorf("function %s already in log space", fn.Name)
		}
		nScope += len(fn.Vars)
		nTab += len(fn.Table)
	}
	const i32, f64 = 4, 8
	card, fixed := newCArray(n*i32), newCArray(n*i32)
	scopeOff, tabOff := newCArray((f+1)*i32), newCArray((f+1)*i32)
	scope, table := newCArray(nScope*i32), newCArray(nTab*f64)
	defer card.free()
	defer fixed.free()
	defer scopeOff.free()
	defer tabOff.free()
	defer scope.free()
	defer table.free()
	// Go views of the C arrays (lengths clamped to >= 1 allocation above; never indexed past n)
	cardS := unsafe.Slice((*C.int32_t)(card.p), n)
	fixedS := unsafe.Slice((*C.int32_t)(fixed.p), n)
	scopeOffS := unsafe.Slice((*C.int32_t)(scopeOff.p), f+1)
	tabOffS := unsafe.Slice((*C.int32_t)(tabOff.p), f+1)
	scopeS := unsafe.Slice((*C.int32_t)(scope.p), nScope)
	tableS := unsafe.Slice((*C.double)(table.p), nTab)
	for i, v := range m.Vars {
		cardS[i], fixedS[i] = C.int32_t(v.Card), C.int32_t(v.FixedVal)
	}
	so, to := 0, 0
	scopeOffS[0], tabOffS[0] = 0, 0
	for i, fn := range m.Funcs {
		for _, v := range fn.Vars {
			scopeS[so] = C.int32_t(v.ID)
			so++
		}
		for _, t := range fn.Table {
			tableS[to] = C.double(t)
			to++
		}
		scopeOffS[i+1], tabOffS[i+1] = C.int32_t(so), C.int32_t(to)
	}
	// the description struct itself is C memory too: it holds nothing but C pointers
	dm := newCArray(int(unsafe.Sizeof(C.gr_model_desc{})))
	defer dm.free()
	d := (*C.gr_model_desc)(dm.p)
	d.n_vars, d.n_funcs = C.int32_t(n), C.int32_t(f)
	d.card, d.fixed = (*C.int32_t)(card.p), (*C.int32_t)(fixed.p)
	d.scope_off, d.scope_var = (*C.int32_t)(scopeOff.p), (*C.int32_t)(scope.p)
	d.tab_off, d.table = (*C.int32_t)(tabOff.p), (*C.double)(table.p)
	var gm *C.gr_model
	err := call(func() C.int {
		gm = C.gr_model_create(d)
		if gm == nil {
			return -1
		}
		return 0
	})
	if err != nil {
		return nil, err
	}
	return gm, nil
}

// create = gr_sampler_create with the error text read on the same OS thread
func create(gm *C.gr_model, o *C.gr_sampler_opts) (*C.gr_sampler, error) {
	var h *C.gr_sampler
	err := call(func() C.int {
		h = C.gr_sampler_create(gm, o)
		if h == nil {
			return -1
		}
		return 0
	})
	return h, err
}

// ScanFlags is OR-ed into gr_sampler_opts.flags by the constructors below.  The default 0 is the reference's random scan
// (one VarSample draw per 64-chain group and update); C.GR_FLAG_SCAN_PER_CHAIN gives every chain its own draw exactly like
// UniformSampler.VarSample (sampler.go:135-174); C.GR_FLAG_SCAN_SYSTEMATIC selects the systematic (chromatic) scan, which is NOT
// the reference's chain law (a host flag such as --scan systematic would set it).
var ScanFlags uint32

// RcclCalls reports how many ncclAllReduce calls gr_allreduce_counts has completed in this process.
func RcclCalls() uint64 { return uint64(C.gr_rccl_calls()) }

// NewHIPChains = NewGibbsSimple / NewGibbsCollapsed + NewChain for nChains chains.
// variants > 1 builds `variants` collapsed groups (Collapse(-1) in each).
func NewHIPChains(m *model.Model, nChains, variants, device int, seed int64, burnIn int64, collapse bool) (*HIPChains, error) {
	gm, err := flatten(m)
	if err != nil {
		return nil, err
	}
	defer C.gr_model_destroy(gm)
	o := C.gr_sampler_opts{
		struct_size: C.uint32_t(unsafe.Sizeof(C.gr_sampler_opts{})),
		n_chains: C.uint32_t(nChains), n_variants: C.uint32_t(variants),
		device: C.int32_t(device), seed: C.uint64_t(seed), flags: C.uint32_t(ScanFlags),
	}
	h, err := create(gm, &o)
	if err != nil {
		return nil, err
	}
	if collapse {
		for k := 0; k < variants; k++ {
			var cv C.int32_t
			if err := call(func() C.int { return C.gr_collapse(h, C.uint32_t(k), -1, &cv, nil) }); err != nil {
				C.gr_sampler_destroy(h)
				return nil, errors.Wrap(err, "Could not collapse random var on startup")
			}
		}
	}
	if err := call(func() C.int { return C.gr_burnin(h, C.uint64_t(burnIn)) }); err != nil {
		C.gr_sampler_destroy(h)
		return nil, errors.Wrap(err, "Failure during chain burn in")
	}
	return &HIPChains{h: h, base: m, nChains: nChains, variants: variants}, nil
}

// AdvanceChain = for every chain: AdvanceChain + wg.Wait with the reference's stopping rule
// (chain.go:184-215) for ConvergenceWindow cw; returns the updates each chain ran.
func (c *HIPChains) AdvanceChain(cw int) (int64, error) {
	var ran C.uint64_t
	if err := call(func() C.int { return C.gr_advance_chain(c.h, C.uint32_t(cw), &ran) }); err != nil {
		return 0, err
	}
	return int64(ran), nil
}

// Advance = for every chain: AdvanceChain + wg.Wait, with an explicit length.
func (c *HIPChains) Advance(updatesPerChain int64) error {
	return call(func() C.int { return C.gr_advance(c.h, C.uint64_t(updatesPerChain)) })
}

func (c *HIPChains) TotalSampleCount() int64 {
	var n C.uint64_t
	if call(func() C.int { return C.gr_total_samples(c.h, &n) }) != nil {
		return 0
	}
	return int64(n)
}

// Merge = MergeChains: fills clones of the base model's variables with the
// un-normalised marginals (prior + counts; collapsed wins).
func (c *HIPChains) Merge() ([]*model.Variable, error) {
	total := 0
	for _, v := range c.base.Vars {
		total += v.Card
	}
	// plain numeric Go slices may be passed to C directly (no Go pointers inside); the handle
	// array must not be: a []*C.gr_sampler holds C pointers only, which cgo allows.
	marg := make([]C.double, total)
	coll := make([]C.int32_t, len(c.base.Vars))
	hs := (**C.gr_sampler)(C.malloc(C.size_t(unsafe.Sizeof(c.h))))
	defer C.free(unsafe.Pointer(hs))
	*hs = c.h
	if err := call(func() C.int { return C.gr_merge(hs, 1, &marg[0], &coll[0]) }); err != nil {
		return nil, err
	}
	out := make([]*model.Variable, len(c.base.Vars))
	off := 0
	for i, v := range c.base.Vars {
		cp := v.Clone()
		for k := 0; k < v.Card; k++ {
			cp.Marginal[k] = float64(marg[off+k])
		}
		cp.Collapsed = coll[i] != 0
		out[i] = cp
		off += v.Card
	}
	return out, nil
}

// Sample satisfies FullSampler by single-stepping chain 0 (debug / compatibility only).
func (c *HIPChains) Sample(s []int) (int, error) {
	if len(s) != len(c.base.Vars) {
		return -1, errors.Errorf("Sample size %d != Var size %d", len(s), len(c.base.Vars))
	}
	if len(s) == 0 {
		return -1, errors.New("empty sample")
	}
	before := make([]C.int32_t, len(s))
	after := make([]C.int32_t, len(s))
	if err := call(func() C.int { return C.gr_state(c.h, 0, &before[0]) }); err != nil {
		return -1, err
	}
	if err := call(func() C.int { return C.gr_advance(c.h, 1) }); err != nil {
		return -1, err
	}
	if err := call(func() C.int { return C.gr_state(c.h, 0, &after[0]) }); err != nil {
		return -1, err
	}
	idx := -1
	for i := range s {
		s[i] = int(after[i])
		if after[i] != before[i] {
			idx = i
		}
	}
	return idx, nil // idx == -1 when the update re-drew the same value
}

// NewHIPAdaptive = `--sampler adaptive` (cmd/root.go:408-415): room for maxChains chain groups
// (reference MaxChains = 128, adaptive.go:49), baseCount of them running un-collapsed from the start.
func NewHIPAdaptive(m *model.Model, chainsPerGroup, maxChains, baseCount, device int, seed int64, burnIn int64) (*HIPChains, error) {
	gm, err := flatten(m)
	if err != nil {
		return nil, err
	}
	defer C.gr_model_destroy(gm)
	o := C.gr_sampler_opts{
		struct_size: C.uint32_t(unsafe.Sizeof(C.gr_sampler_opts{})),
		n_chains: C.uint32_t(chainsPerGroup * maxChains), n_variants: C.uint32_t(maxChains),
		device: C.int32_t(device), seed: C.uint64_t(seed),
	}
	h, err := create(gm, &o)
	if err != nil {
		return nil, err
	}
	if err := call(func() C.int { return C.gr_set_active_groups(h, C.uint32_t(baseCount)) }); err != nil {
		C.gr_sampler_destroy(h)
		return nil, err
	}
	if err := call(func() C.int { return C.gr_burnin(h, C.uint64_t(burnIn)) }); err != nil {
		C.gr_sampler_destroy(h)
		return nil, errors.Wrap(err, "Failure during chain burn in")
	}
	return &HIPChains{h: h, base: m, nChains: chainsPerGroup * maxChains, variants: maxChains}, nil
}

// Adapt = ConvergenceSampler.Adapt (adaptive.go:57-157) after an AdvanceChain: merges the running
// groups, ranks the candidates by ChainConvergence and starts up to newChainCount more groups, each
// with one more collapsed variable. Returns the variables collapsed by the groups it added.
func (c *HIPChains) Adapt(newChainCount int) ([]int, error) {
	if newChainCount < 1 {
		return nil, nil
	}
	vars := make([]C.int32_t, newChainCount)
	var n C.uint32_t
	if err := call(func() C.int { return C.gr_adapt(c.h, C.uint32_t(newChainCount), &vars[0], &n, nil) }); err != nil {
		return nil, errors.Wrap(err, "Could not adapt chains")
	}
	out := make([]int, int(n))
	for i := range out {
		out[i] = int(vars[i])
	}
	return out, nil
}

// ChainCount = len(chains) in cmd/root.go: the groups that currently run.
func (c *HIPChains) ChainCount() int { return int(C.gr_active_groups(c.h)) }

// Close releases the device memory; the handle must not be used afterwards.
func (c *HIPChains) Close() {
	if c.h != nil {
		C.gr_sampler_destroy(c.h)
		c.h = nil
	}
}
