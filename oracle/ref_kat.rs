// ref_kat.rs — generator of the known-answer vectors the reference does not ship (TEST INFRASTRUCTURE, not compiled here:
// this image has no Rust toolchain).  The oracle (oracle/vs_oracle.cpp) restates three things from reading alone — the f32
// expression order of SbqQuantizer::add_sample / quantize, the array mechanics of std::collections::BinaryHeap that decide
// the order of equal-distance candidates, and the rkyv 0.7 byte layout of an archived SbqNode — and no test of the reference
// pins any of them bit for bit.  A maintainer with the reference's toolchain closes that gap with ONE
// command: append the four modules below to the files named in their headers (they need the private items of those
// modules), then
//
//     cargo test --lib ref_kat -- --nocapture --test-threads=1 | grep '^KAT ' > <this repo>/tests/golden/ref_kat.txt
//
// tests/test_oracle_kat.py::test_reference_generated_kats picks the file up and compares every line with the oracle (it is
// skipped while the file is absent).  Inputs are closed-form, so both sides build them without sharing a random generator.

// ====================================================================================================================
// 1. append to pgvectorscale/src/access_method/sbq/quantize.rs      (SbqQuantizer has a private field)
// ====================================================================================================================
#[cfg(test)]
mod ref_kat {
    use super::*;

    fn hex32(v: &[f32]) -> String {
        v.iter().map(|x| format!("{:08x}", x.to_bits())).collect::<Vec<_>>().join(",")
    }
    fn hex64(v: &[u64]) -> String {
        v.iter().map(|x| format!("{:016x}", x)).collect::<Vec<_>>().join(",")
    }
    // sample(i, d) = ((31 i + 17 d) mod 97) / 97 - 0.5, every step in f32
    fn sample(i: usize, d: usize) -> f32 {
        (((i * 31 + d * 17) % 97) as f32) / 97.0f32 - 0.5f32
    }

    fn run(dims: usize, bits: u8, n_train: usize) {
        let mut q = SbqQuantizer {
            use_mean: true,
            training: true,
            count: 0,
            mean: vec![0.0; dims],
            m2: if bits > 1 { vec![0.0; dims] } else { vec![] },
            num_bits_per_dimension: bits,
        };
        for i in 0..n_train {
            let v: Vec<f32> = (0..dims).map(|d| sample(i, d)).collect();
            q.add_sample(&v);
        }
        q.finish_training();
        println!(
            "KAT train dims={} bits={} n={} count={} mean={} m2={}",
            dims, bits, n_train, q.count, hex32(&q.mean), hex32(&q.m2)
        );
        for j in 0..5usize {
            let v: Vec<f32> = (0..dims).map(|d| sample(1000 + 7 * j, d) * 1.5f32).collect();
            let c = q.quantize(&v);
            println!("KAT code dims={} bits={} n={} q={} words={}", dims, bits, n_train, j, hex64(&c));
        }
    }

    #[test]
    fn ref_kat_quantize() {
        run(10, 2, 50);
        run(70, 1, 33);
        run(50, 3, 64);
        run(768, 2, 40);
    }

    // ---- std::collections::BinaryHeap: pop order of equal keys (ListSearchResult.candidates is a
    // BinaryHeap<Reverse<ListSearchNeighbor>> whose Ord looks at the distance only for query scans,
    // AM/graph/neighbor_with_distance.rs:31-43,74-83) ----
    #[derive(Clone, Copy)]
    struct E {
        key: u32,
        id: u32,
    }
    impl PartialEq for E {
        fn eq(&self, o: &Self) -> bool {
            self.key == o.key
        }
    }
    impl Eq for E {}
    impl PartialOrd for E {
        fn partial_cmp(&self, o: &Self) -> Option<std::cmp::Ordering> {
            Some(self.cmp(o))
        }
    }
    impl Ord for E {
        fn cmp(&self, o: &Self) -> std::cmp::Ordering {
            self.key.cmp(&o.key)
        }
    }

    #[test]
    fn ref_kat_binary_heap_tie_order() {
        use std::cmp::Reverse;
        use std::collections::BinaryHeap;
        for (seed, steps, nkeys) in [(1u32, 400usize, 7u32), (2, 3000, 3), (3, 3000, 64)] {
            let mut h: BinaryHeap<Reverse<E>> = BinaryHeap::new();
            let mut x: u32 = seed;
            let mut pops: Vec<String> = vec![];
            for i in 0..steps {
                x = x.wrapping_mul(1103515245).wrapping_add(12345) & 0x7fff_ffff;
                if (x >> 16) % 3 != 0 || h.is_empty() {
                    h.push(Reverse(E { key: (x >> 8) % nkeys, id: i as u32 }));
                } else {
                    pops.push(format!("{}", h.pop().unwrap().0.id));
                }
            }
            while let Some(Reverse(e)) = h.pop() {
                pops.push(format!("{}", e.id));
            }
            println!("KAT heap seed={} steps={} nkeys={} pops={}", seed, steps, nkeys, pops.join(","));
        }
    }
}

// ====================================================================================================================
// 2. append to pgvectorscale/src/access_method/sbq/node.rs          (SbqNode::new and the node fields are private)
// ====================================================================================================================
#[cfg(test)]
mod ref_kat_node {
    use super::*;

    fn hex(b: &[u8]) -> String {
        b.iter().map(|x| format!("{:02x}", x)).collect::<Vec<_>>().join("")
    }

    #[test]
    fn ref_kat_archived_node_bytes() {
        let code: Vec<u64> = vec![0x0123456789abcdef, 0xfedcba9876543210, 0x00000000ffffffff];
        let heap = HeapPointer::new(7, 3);
        // classic node, num_neighbors = 4, two neighbors set
        if let SbqNode::Classic(mut n) = SbqNode::new(heap, 4, false, &code, None) {
            n.neighbor_index_pointers[0] = ItemPointer::new(1, 1);
            n.neighbor_index_pointers[1] = ItemPointer::new(2, 5);
            let bytes = rkyv::to_bytes::<_, 256>(&n).unwrap();
            println!(
                "KAT node kind=classic size={} off_heap={} off_code={} off_nbrs={} off_last={} bytes={}",
                std::mem::size_of::<ArchivedClassicSbqNode>(),
                std::mem::offset_of!(ArchivedClassicSbqNode, heap_item_pointer),
                std::mem::offset_of!(ArchivedClassicSbqNode, bq_vector),
                std::mem::offset_of!(ArchivedClassicSbqNode, neighbor_index_pointers),
                std::mem::offset_of!(ArchivedClassicSbqNode, _neighbor_vectors),
                hex(&bytes)
            );
        }
        // labeled node, labels {2, 5, 9}
        let labels: LabelSet = vec![9i16, 2, 5].into_iter().collect();
        if let SbqNode::Labeled(mut n) = SbqNode::new(heap, 4, true, &code, Some(labels)) {
            n.neighbor_index_pointers[0] = ItemPointer::new(1, 1);
            n.neighbor_index_pointers[1] = ItemPointer::new(2, 5);
            let bytes = rkyv::to_bytes::<_, 256>(&n).unwrap();
            println!(
                "KAT node kind=labeled size={} off_heap={} off_code={} off_nbrs={} off_last={} bytes={}",
                std::mem::size_of::<ArchivedLabeledSbqNode>(),
                std::mem::offset_of!(ArchivedLabeledSbqNode, heap_item_pointer),
                std::mem::offset_of!(ArchivedLabeledSbqNode, bq_vector),
                std::mem::offset_of!(ArchivedLabeledSbqNode, neighbor_index_pointers),
                std::mem::offset_of!(ArchivedLabeledSbqNode, labels),
                hex(&bytes)
            );
        }
    }
}

// ====================================================================================================================
// 3. append to pgvectorscale/src/access_method/distance/distance_x86.rs   (lane order of simdeez's horizontal_add_ps)
// ====================================================================================================================
#[cfg(test)]
mod ref_kat_distance {
    #[test]
    fn ref_kat_simd_bits() {
        // ramps with a non-trivial rounding pattern; the oracle replays the AVX2 accumulation order and must hit these bits
        for d in [8usize, 37, 768, 1536, 2000] {
            let a: Vec<f32> = (0..d).map(|i| ((i * 37 % 101) as f32) / 101.0f32 - 0.3f32).collect();
            let b: Vec<f32> = (0..d).map(|i| ((i * 53 % 103) as f32) / 103.0f32 - 0.7f32).collect();
            let l2 = unsafe { super::distance_l2_x86_avx2(&a, &b) };
            let ip = unsafe { super::inner_product_x86_avx2(&a, &b) };
            println!("KAT simd d={} l2={:08x} ip={:08x}", d, l2.to_bits(), ip.to_bits());
        }
    }
}

// ====================================================================================================================
// 4. append to pgvectorscale/src/access_method/meta_page.rs         (the MetaPage fields are private)
//    The archived MetaPage (String, Option<StartNodes> with its BTreeMap) as rkyv 0.7 really lays it out, with the field
//    offsets of ArchivedMetaPage — what vs_meta_page_decode / oracle/pages_py.py::rkyv_meta_page restate from reading.
// ====================================================================================================================
#[cfg(test)]
mod ref_kat_meta {
    use super::*;

    fn hex(b: &[u8]) -> String {
        b.iter().map(|x| format!("{:02x}", x)).collect::<Vec<_>>().join("")
    }

    fn emit(tag: &str, m: &MetaPage) {
        let bytes = m.serialize_to_vec();
        println!(
            "KAT meta case={} size={} offs={},{},{},{},{},{},{},{},{},{},{},{},{},{} bytes={}",
            tag,
            std::mem::size_of::<ArchivedMetaPage>(),
            std::mem::offset_of!(ArchivedMetaPage, magic_number),
            std::mem::offset_of!(ArchivedMetaPage, version),
            std::mem::offset_of!(ArchivedMetaPage, extension_version_when_built),
            std::mem::offset_of!(ArchivedMetaPage, distance_type),
            std::mem::offset_of!(ArchivedMetaPage, num_dimensions),
            std::mem::offset_of!(ArchivedMetaPage, num_dimensions_to_index),
            std::mem::offset_of!(ArchivedMetaPage, bq_num_bits_per_dimension),
            std::mem::offset_of!(ArchivedMetaPage, storage_type),
            std::mem::offset_of!(ArchivedMetaPage, num_neighbors),
            std::mem::offset_of!(ArchivedMetaPage, search_list_size),
            std::mem::offset_of!(ArchivedMetaPage, max_alpha),
            std::mem::offset_of!(ArchivedMetaPage, start_nodes),
            std::mem::offset_of!(ArchivedMetaPage, quantizer_metadata),
            std::mem::offset_of!(ArchivedMetaPage, has_labels),
            hex(&bytes)
        );
    }

    fn base(version: &str) -> MetaPage {
        MetaPage {
            magic_number: TSV_MAGIC_NUMBER,
            version: TSV_VERSION,
            extension_version_when_built: version.to_string(),
            distance_type: 1,
            num_dimensions: 768,
            num_dimensions_to_index: 512,
            bq_num_bits_per_dimension: 2,
            storage_type: 2,
            num_neighbors: 50,
            search_list_size: 100,
            max_alpha: 1.2,
            start_nodes: None,
            quantizer_metadata: ItemPointer::new(3, 1),
            has_labels: false,
        }
    }

    #[test]
    fn ref_kat_archived_meta_page_bytes() {
        emit("none_inline", &base("0.8.0"));
        let mut m = base("0.8.0-rc1+build.77");
        m.start_nodes = Some(StartNodes::new(ItemPointer::new(7, 1)));
        emit("some_empty_outofline", &m);
        let mut m = base("0.8.0");
        let mut sn = StartNodes::new(ItemPointer::new(7, 1));
        for l in [5i16, -3, 300, 17] {
            sn.upsert(l, ItemPointer::new(100 + l as i32 as u32 % 50, 1 + (l as i32 as u32 % 7) as u16));
        }
        m.start_nodes = Some(sn);
        m.has_labels = true;
        emit("some_four_labels", &m);
        // 1000 labeled start nodes: the B-tree has more than one leaf (4096-byte nodes) and an inner root
        let mut m = base("0.8.0");
        let mut sn = StartNodes::new(ItemPointer::new(7, 1));
        for l in -500i16..500 {
            sn.upsert(l, ItemPointer::new((l as i32 + 1000) as u32, 1 + ((l as i32 + 500) % 90) as u16));
        }
        m.start_nodes = Some(sn);
        m.has_labels = true;
        emit("some_thousand_labels", &m);
    }
}
