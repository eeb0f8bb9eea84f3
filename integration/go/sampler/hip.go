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

func lastErr() error { return errors.New(C.GoString(C.gr_last_error())) }

// flatten copies the Go model into the C description; C memory only, so the
// cgo pointer rules hold (the library never keeps these pointers).
func flatten(m *model.Model) (*C.gr_model, error) {
	n, f := len(m.Vars), len(m.Funcs)
	card := make([]C.int32_t, n)
	fixed := make([]C.int32_t, n)
	for i, v := range m.Vars {
		card[i], fixed[i] = C.int32_t(v.Card), C.int32_t(v.FixedVal)
	}
	scopeOff := make([]C.int32_t, f+1)
	tabOff := make([]C.int32_t, f+1)
	var scope []C.int32_t
	var table []C.double
	for i, fn := range m.Funcs {
		if fn.IsLog {
			return nil, errors.Errorf("function %s already in log space", fn.Name)
		}
		for _, v := range fn.Vars {
			scope = append(scope, C.int32_t(v.ID))
		}
		for _, t := range fn.Table {
			table = append(table, C.double(t))
		}
		scopeOff[i+1], tabOff[i+1] = C.int32_t(len(scope)), C.int32_t(len(table))
	}
	d := C.gr_model_desc{
		n_vars: C.int32_t(n), card: &card[0], fixed: &fixed[0], n_funcs: C.int32_t(f),
		scope_off: &scopeOff[0], scope_var: &scope[0], tab_off: &tabOff[0], table: &table[0],
	}
	gm := C.gr_model_create(&d)
	if gm == nil {
		return nil, lastErr()
	}
	return gm, nil
}

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
		device: C.int32_t(device), seed: C.uint64_t(seed),
	}
	h := C.gr_sampler_create(gm, &o)
	if h == nil {
		return nil, lastErr()
	}
	c := &HIPChains{h: h, base: m, nChains: nChains, variants: variants}
	if collapse {
		for k := 0; k < variants; k++ {
			var cv C.int32_t
			if C.gr_collapse(h, C.uint32_t(k), -1, &cv, nil) != 0 {
				return nil, errors.Wrap(lastErr(), "Could not collapse random var on startup")
			}
		}
	}
	if C.gr_burnin(h, C.uint64_t(burnIn)) != 0 {
		return nil, errors.Wrap(lastErr(), "Failure during chain burn in")
	}
	return c, nil
}

// AdvanceChain = for every chain: AdvanceChain + wg.Wait with the reference's stopping rule
// (chain.go:184-215) for ConvergenceWindow cw; returns the updates each chain ran.
func (c *HIPChains) AdvanceChain(cw int) (int64, error) {
	var ran C.uint64_t
	if C.gr_advance_chain(c.h, C.uint32_t(cw), &ran) != 0 {
		return 0, lastErr()
	}
	return int64(ran), nil
}

// Advance = for every chain: AdvanceChain + wg.Wait, with an explicit length.
func (c *HIPChains) Advance(updatesPerChain int64) error {
	if C.gr_advance(c.h, C.uint64_t(updatesPerChain)) != 0 {
		return lastErr()
	}
	return nil
}

func (c *HIPChains) TotalSampleCount() int64 {
	var n C.uint64_t
	C.gr_total_samples(c.h, &n)
	return int64(n)
}

// Merge = MergeChains: fills clones of the base model's variables with the
// un-normalised marginals (prior + counts; collapsed wins).
func (c *HIPChains) Merge() ([]*model.Variable, error) {
	total := 0
	for _, v := range c.base.Vars {
		total += v.Card
	}
	marg := make([]C.double, total)
	coll := make([]C.int32_t, len(c.base.Vars))
	hs := []*C.gr_sampler{c.h}
	if C.gr_merge(&hs[0], 1, &marg[0], &coll[0]) != 0 {
		return nil, lastErr()
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
	before := make([]C.int32_t, len(s))
	after := make([]C.int32_t, len(s))
	C.gr_state(c.h, 0, &before[0])
	if C.gr_advance(c.h, 1) != 0 {
		return -1, lastErr()
	}
	C.gr_state(c.h, 0, &after[0])
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
	h := C.gr_sampler_create(gm, &o)
	if h == nil {
		return nil, lastErr()
	}
	if C.gr_set_active_groups(h, C.uint32_t(baseCount)) != 0 {
		return nil, lastErr()
	}
	if C.gr_burnin(h, C.uint64_t(burnIn)) != 0 {
		return nil, errors.Wrap(lastErr(), "Failure during chain burn in")
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
	if C.gr_adapt(c.h, C.uint32_t(newChainCount), &vars[0], &n, nil) != 0 {
		return nil, errors.Wrap(lastErr(), "Could not adapt chains")
	}
	out := make([]int, int(n))
	for i := range out {
		out[i] = int(vars[i])
	}
	return out, nil
}

// ChainCount = len(chains) in cmd/root.go: the groups that currently run.
func (c *HIPChains) ChainCount() int { return int(C.gr_active_groups(c.h)) }

func (c *HIPChains) Close() { C.gr_sampler_destroy(c.h) }
