--TEST--
ahocorasick_match_batch: one call over several haystacks equals a loop of ahocorasick_match calls
--SKIPIF--
<?php if (!extension_loaded("ahocorasick")) print "skip"; ?>
--FILE--
<?php
$patterns = array(
    array('key' => 'ab', 'value' => 'alfa'),
    array('id' => 7, 'value' => 'lfa'),
    array('value' => 'omega', 'aux' => array(1, 2)),
    array('value' => 'a'),
);
$c = ahocorasick_init($patterns);
$haystacks = array('alfa omega', '', 'no hit here? a!', str_repeat('x', 9000) . 'omegalfa');
$batch = ahocorasick_match_batch($haystacks, $c);
$loop = array();
foreach ($haystacks as $h) {
    $loop[] = ahocorasick_match($h, $c);
}
var_dump($batch === $loop);
var_dump(count($batch));
var_dump($batch[0][0]);
var_dump(count($batch[1]));
$first = ahocorasick_match_batch($haystacks, $c, false);
var_dump($first[0] === ahocorasick_match($haystacks[0], $c, false));
var_dump(ahocorasick_match_batch(array(), $c));
ahocorasick_deinit($c);
?>
--EXPECT--
bool(true)
int(4)
array(3) {
  ["pos"]=>
  int(1)
  ["start_postion"]=>
  int(0)
  ["value"]=>
  string(1) "a"
}
int(0)
bool(true)
array(0) {
}
