dnl phpize build of the Zend shim against a prebuilt libacmi.so (not run in this repo's image).
dnl   cd php_ext && phpize && ./configure --enable-ahocorasick --with-acmi=/path/to/repo && make && make test
PHP_ARG_ENABLE(ahocorasick, whether to enable the MI355X ahocorasick engine, [  --enable-ahocorasick   Enable ahocorasick])
PHP_ARG_WITH(acmi, repo root holding include/acmi.h and php_aho_corasick_amd/libacmi.so, [  --with-acmi=DIR], ..)
if test "$PHP_AHOCORASICK" != "no"; then
  PHP_ADD_INCLUDE($PHP_ACMI/include)
  PHP_ADD_LIBRARY_WITH_PATH(acmi, $PHP_ACMI/php_aho_corasick_amd, AHOCORASICK_SHARED_LIBADD)
  PHP_SUBST(AHOCORASICK_SHARED_LIBADD)
  PHP_NEW_EXTENSION(ahocorasick, php_ahocorasick_amd.c, $ext_shared)
fi
